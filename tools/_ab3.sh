set -e
L=gpurun_out/partdbg.log
echo "== LT_PART_DEBUG=2: ranking on, contiguous write-out" > $L
timeout -k 10 200 python - >> $L 2>&1 <<'PY'
import sys; sys.path.insert(0,'.')
import light_transport_amd as lt
from tests import scenes as S
ctx = lt.Context(0)
prob = S.slab(n=256, voxel=0.1); prob.apply(ctx, "f64"); ctx.set_tally_mode("log"); ctx.set_overlap(1)
for r in range(4):
    ctx.zero_tally(); ctx.launch(10**7, seed=r); ctx.sync()
    print(ctx.last_kernel_ms(), ctx.last_log_stages(), flush=True)
PY
cat $L
