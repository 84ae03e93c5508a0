import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import light_transport_amd as lt
from tests import scenes as S
big = S.Problem([(0.1, 10.0, 0.9, 1.0)], (300, 300, 200), (-15.0, -15.0, 0.0), (0.1,) * 3,
                layers=dict(z_bounds=[0.0, np.inf], medium_idx=[0]))
ctx = lt.Context(0)
for name, prob, n in (("big", big, 300000), ("slab_b3", S.slab(), 300000), ("c5geom", S.two_layer(n=512, voxel=0.025), 2000000)):
    ref = None
    for mode, hot, lanes in (("atomic", None, 1), ("log", "0", 1), ("log", None, 1), ("log", "7", 1), ("log", None, 2), ("log", "100000", 2)):
        knobs = {}
        if hot is not None: knobs["log_hot"] = int(hot)
        if name == "slab_b3": knobs["log_bits2"] = 3
        with ctx.tuning(**knobs):       # (the library reads the environment only when a context is created: lt_set_tuning)
            prob.apply(ctx, "u64fx"); ctx.set_tally_mode(mode); ctx.set_overlap(lanes)
            ctx.launch(n, seed=5); ctx.sync()
        g, c = ctx.read_grid_raw(), ctx.read_counters()
        info = ctx.last_log_info()
        ht = ctx.last_log_hot_tiles() if mode == "log" else None
        if ref is None: ref = (g, c)
        ok = np.array_equal(g, ref[0]) and c["steps"] == ref[1]["steps"]
        print(name, mode, "log_hot", hot, "lanes", lanes, "hot tiles", ht, info, "OK" if ok else "MISMATCH %d voxels" % int((g != ref[0]).sum()), flush=True)
        if not ok: sys.exit(1)
ctx.close()
