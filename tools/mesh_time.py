"""Mesh walks beyond the LDS budget (tables in global memory): 5140-triangle sphere in a box, f64 / f32 walk, log tally,
with the near-triangle lists of the clearance records (default), without them (LT_NO_NEAR_LISTS=1) and without the
clearance grid (LT_NO_CLEARANCE=1).    python tools/mesh_time.py [photons]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import light_transport_amd as lt
from tests import scenes as S

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2 * 10 ** 6
prob = S.sphere_in_box(4, split_method=0)[0]
if os.environ.get("LT_MESH_DENSE_MEDIA"):      # 10x the scattering: mean free path 0.02, surface queries become rare
    prob.media = [(0.05, 50.0, 0.8, 1.0), (0.8, 80.0, 0.9, 1.37)]
for env in ({}, {"LT_NO_NEAR_LISTS": "1"}, {"LT_NO_CLEARANCE": "1"}):
    os.environ.update(env)
    ctx = lt.Context(0)
    for dtype, f32 in (("f64", False), ("f32", True)):
        prob.apply(ctx, dtype); ctx.set_tally_mode("log"); ctx.set_overlap(1)
        best = 1e9
        for r in range(3):
            ctx.zero_tally(); ctx.launch(n, seed=r, f32_walk=f32); ctx.sync()
            if r: best = min(best, ctx.last_log_stages()["walk_ms"])
        c = ctx.read_counters()
        print("sphere 5140 tris, %s walk, %-22s walk %7.2f ms  %6.2f Gsteps/s (%.1f steps/photon)" % (
            dtype, ",".join(env) or "near lists + clearance", best, c["steps"] / best / 1e6, c["steps"] / n), flush=True)
    ctx.close()
    for k in env: os.environ.pop(k, None)
