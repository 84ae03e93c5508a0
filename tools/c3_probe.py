"""One C3 job (two-layer skin model, Fresnel interfaces, 1e7 photons) -- target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
prob = S.two_layer(n=256, voxel=0.05)
prob.apply(ctx, "f64")
for r in range(2):
    ctx.zero_tally(); ctx.launch(10 ** 7, seed=r); ctx.sync()
print("%.2f ms, %d steps" % (ctx.last_kernel_ms(), ctx.read_counters()["steps"]), ctx.last_log_stages())
