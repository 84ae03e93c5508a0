"""How much of the walk is drain tail?  Walk-only timing (LT_DIAG_NO_TALLY=1) at several photon counts: the tail is the
intercept of t(n)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
S.slab(n=256, voxel=0.1).apply(ctx, "f64"); ctx.set_tally_mode("atomic")
ts = {}
for n in (1250000, 2500000, 5000000, 10000000, 20000000):
    best = 1e9
    for r in range(3):
        ctx.zero_tally(); ctx.launch(n, seed=r); ctx.sync(); best = min(best, ctx.last_kernel_ms())
    ts[n] = best
    print(n, "%.2f ms" % best, flush=True)
ns = np.array(sorted(ts)); t = np.array([ts[k] for k in ns])
a, b = np.polyfit(ns, t, 1)
print("fit: %.3f ms per 1e6 photons + %.2f ms intercept (tail + launch)" % (a * 1e6, b))
