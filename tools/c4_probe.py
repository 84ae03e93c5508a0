"""One C4 job (Cornell cavity + cone mesh, 1e7 photons, atomic tally) -- target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
prob = S.cornell(256)
prob.apply(ctx, "f64"); ctx.set_tally_mode("atomic")
for r in range(2):
    ctx.zero_tally(); ctx.launch(10 ** 7, seed=r); ctx.sync()
print("%.2f ms, %d steps" % (ctx.last_kernel_ms(), ctx.read_counters()["steps"]))
