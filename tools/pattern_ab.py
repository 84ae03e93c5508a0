"""A/B of the sub-batch layout of an overlapped launch (lt_set_overlap 2), compared by median device time.
    python tools/pattern_ab.py [c2|c5] [reps] [lanes:PATTERN=2,2,1 ...]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import light_transport_amd as lt
from tests import scenes as S

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
prob, n = {"c2": (S.slab(n=256, voxel=0.1), 10 ** 7), "c5": (S.two_layer(n=512, voxel=0.025), 12500000),
           "c4": (S.cornell(256), 10 ** 7)}[which]
# (lanes, knob, value)
variants = [(2, "PATTERN", "2,2,1"), (3, "PATTERN", "2,2,1"), (3, "PATTERN", "1,1,1"), (3, "PATTERN", "3,3,2"), (3, "PATTERN", "4,4,1"),
            (3, "PATTERN", "2,2,2,1"), (3, "PATTERN", "3,3,1,1"), (2, "PATTERN", "5,5,2")]
if len(sys.argv) > 3:
    variants = [(int(a.split(":")[0]),) + tuple(a.split(":")[1].split("=")) for a in sys.argv[3:]]
# The library reads LT_OVERLAP_PATTERN once, when a context is created (lt_set_tuning holds the numeric knobs; the pattern
# is a string): one context per variant, created, measured and closed in turn -- the variants are NOT interleaved launch by
# launch any more (a C2 context holds ~90 GB of logs; eight of them do not fit side by side), so compare medians.
times = {v: [] for v in variants}
for v in variants:
    os.environ.pop("LT_OVERLAP_PATTERN", None)
    os.environ["LT_OVERLAP_" + v[1]] = v[2]
    ctx = lt.Context(0)
    prob.apply(ctx, "f64"); ctx.set_tally_mode("log"); ctx.set_overlap(v[0])
    for r in range(reps + 1):
        ctx.zero_tally(); ctx.launch(n, seed=r); ctx.sync()
        if r:
            times[v].append(ctx.last_kernel_ms())
    ctx.close()
for v in variants:
    t = times[v]
    print("%s lanes %d %-8s %-9s median %.2f  min %.2f  max %.2f ms" % (which, v[0], v[1], v[2], statistics.median(t), min(t), max(t)), flush=True)
