#!/bin/bash
# GPU box: bench + rocprof kernel trace + PMC traffic passes.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tee gpurun_out/smoke.log &&
python bench.py --steps 5 --warmup 1 2>&1 | tee gpurun_out/bench_n1.log &&
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tee gpurun_out/bench_dist1.log &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_trace.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_write.log 2>&1
echo "rc=$?"
find gpurun_out -name "*.csv" | head -30
