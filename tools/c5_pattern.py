"""C5 share (two-layer, 512^3, 1.25e7 photons), two lanes: device time of one launch for several batch layouts
(LT_OVERLAP_PATTERN is read once per context, in lt_create).   python tools/c5_pattern.py [case] [patterns ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import light_transport_amd as lt
from tests import scenes as S
case = sys.argv[1] if len(sys.argv) > 1 else "c5"
pats = sys.argv[2:] or ["", "5,3,2", "3,2,1", "9,7,4", "2,2,1,1", "3,3,2"]
prob, n = {"c5": (S.two_layer(n=512, voxel=0.025), 12500000), "c2": (S.slab(n=256, voxel=0.1), 10 ** 7),
           "c3": (S.two_layer(n=256, voxel=0.05), 10 ** 7)}[case]
for pat in pats:
    if pat: os.environ["LT_OVERLAP_PATTERN"] = pat
    else: os.environ.pop("LT_OVERLAP_PATTERN", None)
    ctx = lt.Context(0)
    prob.apply(ctx, "f64"); ctx.set_tally_mode("log"); ctx.set_overlap(2)
    t = []
    for r in range(4):
        ctx.zero_tally(); ctx.launch(n, seed=r); ctx.sync()
        t.append(ctx.last_kernel_ms())
    info = ctx.last_log_info()
    print("%s pattern %-10s: %s ms (best %.2f), %d batches" % (case, pat or "default", " ".join("%.2f" % x for x in t[1:]), min(t[1:]), info["batches"]), flush=True)
    ctx.close()
