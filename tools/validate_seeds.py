"""Matched-slab benchmark (van de Hulst via WJZ95 Table 1: Rd 0.09739, Tt 0.66096) over several seeds at 1e8 photons each:
mean and scatter of Rd / Tt -- the check that exposed XORWOW's weak seed-only initialisation, rerun
whenever the way the walk consumes its random stream changes.    python tools/validate_seeds.py [seeds=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
n, k = 10 ** 8, int(sys.argv[1]) if len(sys.argv) > 1 else 8
slab = S.slab(media=((10.0, 90.0, 0.75, 1.0),), thickness=0.02, n=8, voxel=0.0025)
semi = S.slab(media=((10.0, 90.0, 0.0, 1.5),), n=8, voxel=1.0)
rd, tt, rr = [], [], []
for seed in range(1, k + 1):
    slab.apply(ctx, "f64"); ctx.zero_tally(); ctx.launch(n, seed=seed); ctx.sync(); c = ctx.read_counters()
    rd.append(c["w_escaped_top"] / n); tt.append(c["w_escaped_bottom"] / n)
    semi.apply(ctx, "f64"); ctx.zero_tally(); ctx.launch(n, seed=100 + seed); ctx.sync(); c = ctx.read_counters()
    rr.append((c["w_escaped_top"] + c["w_specular"]) / n)
    print("seed %d: Rd %.6f  Tt %.6f | semi-infinite R %.6f" % (seed, rd[-1], tt[-1], rr[-1]), flush=True)
for name, v, ref in (("Rd", rd, 0.09739), ("Tt", tt, 0.66096), ("R (semi-infinite, mismatched)", rr, 0.26000)):
    v = np.array(v)
    print("%s: mean %.6f +- %.6f (standard error of the mean; scatter of one run %.6f)   reference %.5f   deviation %+.1e" % (
        name, v.mean(), v.std(ddof=1) / np.sqrt(len(v)), v.std(ddof=1), ref, v.mean() - ref))
