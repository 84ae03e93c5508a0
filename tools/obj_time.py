"""Walk time on the reference's OBJ assets (fixture G10: teapot / cow / pumpkin, 5.8k-10k triangles, in a closed box;
tables in global memory), f64 and f32 walk, log tally.    python tools/obj_time.py [photons]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g10_obj_meshes.npz"))
ctx = lt.Context(0)
for name in ("teapot", "cow", "pumpkin"):
    prob = S.obj_in_box(g[name + "_verts"], g[name + "_faces"])[0]
    for dtype, f32 in ((("f64", False),) if os.environ.get("OBJ_F64_ONLY") else (("f64", False), ("f32", True))):
        prob.apply(ctx, dtype); ctx.set_tally_mode("log"); ctx.set_overlap(1)
        best = 1e9
        for r in range(3):
            ctx.zero_tally(); ctx.launch(n, seed=r, f32_walk=f32); ctx.sync()
            if r: best = min(best, ctx.last_log_stages()["walk_ms"])
        c = ctx.read_counters()
        print("%-8s %5d triangles, %s walk: %7.2f ms  %6.2f Gsteps/s (%.1f steps/photon)" % (
            name, len(g[name + "_faces"]), dtype, best, c["steps"] / best / 1e6, c["steps"] / n), flush=True)
ctx.close()
