import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
for name, prob in (("c2", S.slab(n=256, voxel=0.1)), ("c3", S.two_layer(n=256, voxel=0.05))):
    prob.apply(ctx, "f64"); ctx.set_tally_quantity("fluence"); ctx.set_tally_mode("log")
    ctx.zero_tally(); ctx.launch(2 * 10 ** 6, seed=1); ctx.sync()
    g = np.asarray(ctx.read_grid()).reshape(256, 256, 256)      # [z][y][x]
    tot = g.sum()
    for bz, by, bx in ((16, 16, 16), (8, 32, 32), (16, 32, 32), (32, 32, 32), (16, 64, 64)):
        # box anchored at z = 0, centred on the beam
        c = 128
        s = g[0:bz, c - by // 2:c + by // 2, c - bx // 2:c + bx // 2].sum()
        print(name, "box z%d y%d x%d (%d voxels, %d KiB f64): %.1f %% of the fluence-weighted deposits" % (bz, by, bx, bz * by * bx, bz * by * bx * 8 // 1024, 100 * s / tot))
ctx.close()
