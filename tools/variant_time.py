import os, sys, glob, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    import shutil
    shutil.copy(sys.argv[1], os.path.join(root, "light_transport_amd", "liblt_hip.so"))
    sys.path.insert(0, root)
    import numpy as np
    import light_transport_amd as lt
    from tests import scenes as S
    ctx = lt.Context(0)
    c2 = S.slab(n=256, voxel=0.1)
    for dtype, f32 in (("f64", False), ("f32", True)):
        c2.apply(ctx, dtype); ctx.set_tally_mode("log", 0)
        ctx.launch(10**6, seed=9, f32_walk=f32); ctx.sync()
        best = 1e9
        for r in range(3):
            ctx.zero_tally(); ctx.launch(10**7, seed=r, f32_walk=f32); ctx.sync(); best = min(best, ctx.last_kernel_ms())
        c = ctx.read_counters()
        print(os.path.basename(sys.argv[1]), dtype, "log total %.2f ms %.2f Gsteps/s" % (best, c["steps"] / best / 1e6), flush=True)
        ctx.set_tally_mode("atomic")
        ctx.zero_tally(); ctx.launch(10**7, seed=1, f32_walk=f32); ctx.sync()
        print(os.path.basename(sys.argv[1]), dtype, "atomic %.2f ms" % ctx.last_kernel_ms(), flush=True)
else:
    keep = os.path.join(root, "light_transport_amd", "liblt_hip.so.keep")
    import shutil
    shutil.copy(os.path.join(root, "light_transport_amd", "liblt_hip.so"), keep)
    for v in sorted(glob.glob(os.path.join(root, "light_transport_amd", "variants", "*.so"))):
        subprocess.call([sys.executable, __file__, v])
    shutil.copy(keep, os.path.join(root, "light_transport_amd", "liblt_hip.so"))
