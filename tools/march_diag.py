"""One launch per OBJ mesh with the instrumented build (make EXTRA_KFLAGS=-DLT_DIAG_MESH): prints the walk's per-wave
counters.    python tools/march_diag.py [photons] [mesh,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2 * 10 ** 6
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["teapot", "pumpkin"]
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g10_obj_meshes.npz"))
ctx = lt.Context(0)
for name in names:
    prob = S.obj_in_box(g[name + "_verts"], g[name + "_faces"])[0]
    prob.apply(ctx, "f64"); ctx.set_tally_mode("log"); ctx.set_overlap(1)
    ctx.zero_tally(); ctx.launch(n, seed=1); ctx.sync()
    print(name, ctx.last_log_stages()["walk_ms"], "ms", flush=True)
ctx.close()
