import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
c2 = S.slab(n=256, voxel=0.1)
for dtype, f32 in (("f64", False), ("u64fx", False), ("f32", False), ("f32", True)):
    c2.apply(ctx, dtype); ctx.set_tally_mode("log", int(os.environ.get("LOG_GB", "0")) << 30)
    for rep in range(2):
        ctx.zero_tally(); ctx.launch(10**7, seed=rep, f32_walk=f32); ctx.sync()
        c = ctx.read_counters(); ms = ctx.last_kernel_ms()
        print("C2 %s %s total %.2f ms  %.2f Gsteps/s" % (dtype, "f32walk" if f32 else "f64walk", ms, c["steps"] / ms / 1e6), flush=True)
