"""C4 (Cornell cavity + cone, BVH) timings: tally mode x precision x lanes.   python tools/c4_time.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
prob, n = S.cornell(256), 10 ** 7
# C4_KNOBS="query_min=1,force_march=1": experiment knobs (lt_set_tuning) for every row
for kv in filter(None, os.environ.get("C4_KNOBS", "").split(",")):
    ctx.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
rows = (("atomic", "f64", False, 1), ("log", "f64", False, 1), ("log", "f64", False, 2), ("log", "f32", True, 1))
if os.environ.get("C4_ROWS"):
    rows = tuple(r for k, r in enumerate(rows) if str(k) in os.environ["C4_ROWS"].split(","))
for mode, dtype, f32, lanes in rows:
    prob.apply(ctx, dtype); ctx.set_tally_mode(mode); ctx.set_overlap(lanes)
    best = 1e9
    for r in range(3):
        ctx.zero_tally(); ctx.launch(n, seed=r, f32_walk=f32); ctx.sync()
        if r: best = min(best, ctx.last_kernel_ms())
    c = ctx.read_counters(); st = ctx.last_log_stages()
    print("C4 %-6s %s walk lanes %d: %7.2f ms  %6.2f Gsteps/s %s" % (mode, "f32" if f32 else "f64", lanes, best, c["steps"] / best / 1e6,
          ("| walk %.2f part %.2f reduce %.2f" % (st["walk_ms"], st["partition_ms"], st["reduce_ms"])) if st else ""), flush=True)
