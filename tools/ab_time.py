import os, sys, glob, subprocess, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    shutil.copy(sys.argv[1], os.path.join(root, "light_transport_amd", "liblt_hip.so"))
    sys.path.insert(0, root)
    import numpy as np
    import light_transport_amd as lt
    from tests import scenes as S
    ctx = lt.Context(0)
    c2 = S.slab(n=256, voxel=0.1)
    c2.apply(ctx, "f64"); ctx.set_tally_mode("log", 0)
    for r in range(4):
        ctx.zero_tally(); ctx.launch(10**7, seed=r); ctx.sync()
        print(os.path.basename(sys.argv[1]), "total %.2f ms" % ctx.last_kernel_ms(), file=sys.stderr, flush=True)
else:
    keep = os.path.join(root, "light_transport_amd", "liblt_hip.so.keep")
    shutil.copy(os.path.join(root, "light_transport_amd", "liblt_hip.so"), keep)
    vs = sorted(glob.glob(os.path.join(root, "light_transport_amd", "variants", "*.so")))
    for v in vs + vs:
        subprocess.call([sys.executable, __file__, v])
    shutil.copy(keep, os.path.join(root, "light_transport_amd", "liblt_hip.so"))
