#!/bin/bash
# GPU box: bench + rocprof kernel trace + PMC passes for the default bench workload.
set -o pipefail
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py 2>&1 | tee gpurun_out/bench_n1.log &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alone > gpurun_out/prof_trace.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_write.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_sq.log 2>&1
echo "rc=$?"
