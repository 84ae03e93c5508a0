"""Diagnostic: which fraction of C2's atomic requests falls into a candidate LDS tile around the source?
Run under rocprofv3 --pmc TCC_EA0_ATOMIC_sum; each launch uses a grid equal to the candidate tile, so only
in-tile deposits issue atomics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
ctx = lt.Context(0)
ctx.set_media([(0.1, 10.0, 0.9, 1.0)]); ctx.set_layers([0.0, np.inf], [0]); ctx.set_source(0, (0, 0, 0), (0, 0, 1))
n = 2000000
tiles = [((256, 256, 256), 0.0), ((32, 32, 16), 0.0), ((16, 16, 64), 0.0), ((24, 24, 28), 0.0), ((24, 24, 28), 0.3),
         ((32, 32, 32), 0.0), ((64, 64, 32), 0.0), ((64, 64, 64), 0.0)]
for shape, z0 in tiles:
    ctx.set_grid(shape, (-shape[0] * 0.05, -shape[1] * 0.05, z0), (0.1,) * 3, "f64")
    ctx.launch(n, seed=1); ctx.sync()
    c = ctx.read_counters()
    print(shape, z0, "weight captured %.3f" % (c["w_absorbed"] / (c["w_absorbed"] + c["w_lost_outside_grid"])), flush=True)
