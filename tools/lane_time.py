"""Device time of ONE lt_launch per regime: lanes 1 / 2 (lt_set_overlap), per configuration, with the per-stage sums.
    python tools/lane_time.py [c2|c3|c5|c2f32|all] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import light_transport_amd as lt
from tests import scenes as S

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cases = {
    "c2": (S.slab(n=256, voxel=0.1), 10 ** 7, "f64", False),
    "c2f32": (S.slab(n=256, voxel=0.1), 10 ** 7, "f32", True),
    "c3": (S.two_layer(n=256, voxel=0.05), 10 ** 7, "f64", False),
    "c5": (S.two_layer(n=512, voxel=0.025), 12500000, "f64", False),
}
ctx = lt.Context(0)
for name in (list(cases) if which == "all" else which.split(",")):
    prob, n, dtype, f32 = cases[name]
    prob.apply(ctx, dtype); ctx.set_tally_mode("log")
    for lanes in (1, 2, 3):
        ctx.set_overlap(lanes)
        best, st_best = 1e9, None
        for r in range(reps + 1):
            ctx.zero_tally(); ctx.launch(n, seed=r, f32_walk=f32); ctx.sync()
            ms = ctx.last_kernel_ms()
            if r and ms < best:
                best, st_best = ms, ctx.last_log_stages()
        c, info = ctx.read_counters(), ctx.last_log_info()
        print("%-6s lanes %d: %7.2f ms  %6.2f Gsteps/s | stage sums: walk %.2f scan %.2f part %.2f reduce %.2f | %d batches, "
              "%.1f rec/photon, overflow %d" % (name, lanes, best, c["steps"] / best / 1e6, st_best["walk_ms"], st_best["scan_ms"],
                                                st_best["partition_ms"], st_best["reduce_ms"], info["batches"],
                                                info["records"] / n, info["overflow_records"]), flush=True)
ctx.close()
