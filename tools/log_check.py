import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
def run(prob, n, dtype, mode, seed=3, log_bytes=0, f32=False):
    prob.apply(ctx, dtype); ctx.set_tally_mode(mode, log_bytes)
    ctx.launch(n, seed=seed, f32_walk=f32); ctx.sync()
    return ctx.read_grid_raw(), ctx.read_counters(), ctx.last_kernel_ms()
for name, prob, n in (("slab64", S.slab(), 200000), ("two_layer", S.two_layer(n=64), 200000), ("cornell", S.cornell(64), 100000),
                      ("slab odd 100x70x33", S.Problem([(0.1, 10.0, 0.9, 1.0)], (100, 70, 33), (-5.0, -3.5, 0.0), (0.1,) * 3,
                                                     layers=dict(z_bounds=[0.0, np.inf], medium_idx=[0])), 200000)):
    a, ca, _ = run(prob, n, "u64fx", "atomic")
    b, cb, _ = run(prob, n, "u64fx", "log")
    c_, cc, _ = run(prob, n, "u64fx", "log", log_bytes=64 << 20)   # tiny log: many batches + overflow fallback
    print(name, "u64fx log==atomic:", np.array_equal(a, b), "tiny-log==atomic:", np.array_equal(a, c_), "steps", ca["steps"] == cb["steps"] == cc["steps"],
          "sum", int(a.sum()) == int(b.sum()), flush=True)
    a, _, _ = run(prob, n, "f64", "atomic"); b, _, _ = run(prob, n, "f64", "log")
    print("   f64 max rel diff", float(np.abs(a - b).max() / a.max()), flush=True)
ctx.set_tally_mode("log", 16 << 30)
c2 = S.slab(n=256, voxel=0.1)
for dtype, f32 in (("f64", False), ("u64fx", False), ("f32", True), ("f64", True)):
    for mode in ("atomic", "log", "log"):
        g, c, ms = run(c2, 10**7, dtype, mode, seed=1, f32=f32)
        print("C2 %-5s walk=%s mode=%-6s %8.2f ms  %6.2f Gsteps/s" % (dtype, "f32" if f32 else "f64", mode, ms, c["steps"] / ms / 1e6), flush=True)
