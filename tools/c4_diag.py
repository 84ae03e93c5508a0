import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
prob = S.cornell(256)
prob.apply(ctx, "f64"); ctx.set_tally_mode("log"); ctx.set_overlap(1)
for r in range(2):
    ctx.zero_tally(); ctx.launch(10 ** 7, seed=r); ctx.sync()
    print("launch", r, ctx.last_kernel_ms(), ctx.last_log_stages(), flush=True)
c = ctx.read_counters(); print(c)
