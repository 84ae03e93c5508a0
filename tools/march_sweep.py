"""Sweep of the march-grid knobs on the reference's OBJ meshes (fixture G10), f64 walk, log tally.
    python tools/march_sweep.py [photons] [mesh,...]      env knobs swept: LT_QUERY_MIN x LT_MARCH_CELLS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import light_transport_amd as lt
from tests import scenes as S

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4 * 10 ** 6
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["teapot", "pumpkin"]
qmins = [int(x) for x in os.environ.get("SWEEP_QMIN", "8,24,48").split(",")]
cells = os.environ.get("SWEEP_CELLS", "64,128,192").split(",")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g10_obj_meshes.npz"))
for name in names:
    prob = S.obj_in_box(g[name + "_verts"], g[name + "_faces"])[0]
    for cl in cells:
        if cl and cl != "auto":
            os.environ["LT_MARCH_CELLS"] = cl
        else:
            os.environ.pop("LT_MARCH_CELLS", None)
        for qm in qmins:
            os.environ["LT_QUERY_MIN"] = str(qm)
            ctx = lt.Context(0)
            prob.apply(ctx, "f64"); ctx.set_tally_mode("log"); ctx.set_overlap(1)
            best = 1e9
            for r in range(3):
                ctx.zero_tally(); ctx.launch(n, seed=r); ctx.sync()
                if r: best = min(best, ctx.last_log_stages()["walk_ms"])
            c = ctx.read_counters()
            print("%-8s cells %-4s query_min %2d: walk %7.2f ms  %6.2f Gsteps/s" % (name, cl, qm, best, c["steps"] / best / 1e6), flush=True)
            ctx.close()
