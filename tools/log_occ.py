import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
c2 = S.slab(n=256, voxel=0.1)
for dtype, f32 in (("f64", False), ("f32", True)):
    c2.apply(ctx, dtype); ctx.set_tally_mode("log", 0)
    ctx.launch(10**6, seed=9, f32_walk=f32); ctx.sync()   # calibrate record rate
    for bpc, thr in ((1, 256), (2, 256), (3, 256), (4, 256), (2, 128), (4, 128), (6, 128), (8, 64)):
        try:
            ctx.set_launch_config(bpc, thr)
            ctx.zero_tally(); ctx.launch(10**7, seed=1, f32_walk=f32); ctx.sync()
            c = ctx.read_counters(); ms = ctx.last_kernel_ms()
            print("C2 %s bpc=%d thr=%d total %.2f ms  %.2f Gsteps/s" % (dtype, bpc, thr, ms, c["steps"] / ms / 1e6), file=sys.stderr, flush=True)
        except Exception as e:
            print("skip", bpc, thr, str(e)[:60], file=sys.stderr)
