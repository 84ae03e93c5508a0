"""Literature known answers on the GPU at 1e8 photons (f64 walk; the f32 walk beside it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
n = 10 ** 8
slab = S.slab(media=((10.0, 90.0, 0.75, 1.0),), thickness=0.02, n=8, voxel=0.0025)
semi = S.slab(media=((10.0, 90.0, 0.0, 1.5),), n=8, voxel=1.0)
for f32, tally in ((False, "f64"), (True, "f32")):
    slab.apply(ctx, tally); ctx.zero_tally(); ctx.launch(n, seed=1, f32_walk=f32); ctx.sync(); c = ctx.read_counters()
    print("%s walk, matched slab (van de Hulst via WJZ95 Table 1: Rd 0.09739, Tt 0.66096): Rd %.5f  Tt %.5f  A %.5f" % (
        "f32" if f32 else "f64", c["w_escaped_top"] / n, c["w_escaped_bottom"] / n, c["w_absorbed"] / n), flush=True)
    semi.apply(ctx, tally); ctx.zero_tally(); ctx.launch(n, seed=2, f32_walk=f32); ctx.sync(); c = ctx.read_counters()
    print("%s walk, mismatched semi-infinite isotropic (van de Hulst / Giovanelli: R 0.26000): R %.5f (specular %.5f)" % (
        "f32" if f32 else "f64", (c["w_escaped_top"] + c["w_specular"]) / n, c["w_specular"] / n), flush=True)

# WJZ95 Table 3: three layers (n = 1.37 each, ambient 1): (mu_a, mu_s, g, d) = (1, 100, 0.9, 0.1), (1, 10, 0, 0.1),
# (2, 10, 0.7, 0.2) cm; published Rd 0.2375 (MCML) / 0.2381 (Gardner et al.), Tt 0.0965 / 0.0974 -- Monte Carlo
# results themselves, good to about 1e-3
import numpy as np
three = S.Problem([(1.0, 100.0, 0.9, 1.37), (1.0, 10.0, 0.0, 1.37), (2.0, 10.0, 0.7, 1.37)], (8, 8, 8), (-1.0, -1.0, 0.0),
                  (0.25, 0.25, 0.05), layers=dict(z_bounds=[0.0, 0.1, 0.2, 0.4], medium_idx=[0, 1, 2], n_above=1.0, n_below=1.0))
for f32, tally in ((False, "f64"), (True, "f32")):
    three.apply(ctx, tally); ctx.zero_tally(); ctx.launch(n, seed=3, f32_walk=f32); ctx.sync(); c = ctx.read_counters()
    print("%s walk, three-layer tissue (WJZ95 Table 3: Rd 0.2375 / 0.2381, Tt 0.0965 / 0.0974): Rd %.5f  Tt %.5f  specular %.5f" % (
        "f32" if f32 else "f64", c["w_escaped_top"] / n, c["w_escaped_bottom"] / n, c["w_specular"] / n), flush=True)
