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
        if hot is not None: os.environ["LT_LOG_HOT"] = hot
        else: os.environ.pop("LT_LOG_HOT", None)
        if name == "slab_b3": os.environ["LT_LOG_BITS2"] = "3"
        try:
            prob.apply(ctx, "u64fx"); ctx.set_tally_mode(mode); ctx.set_overlap(lanes)
            ctx.launch(n, seed=5); ctx.sync()
        finally:
            os.environ.pop("LT_LOG_BITS2", None)
        g, c = ctx.read_grid_raw(), ctx.read_counters()
        info = ctx.last_log_info()
        ht = ctx.last_log_hot_tiles() if mode == "log" else None
        if ref is None: ref = (g, c)
        ok = np.array_equal(g, ref[0]) and c["steps"] == ref[1]["steps"]
        print(name, mode, "LT_LOG_HOT", hot, "lanes", lanes, "hot tiles", ht, info, "OK" if ok else "MISMATCH %d voxels" % int((g != ref[0]).sum()), flush=True)
        if not ok: sys.exit(1)
ctx.close()
