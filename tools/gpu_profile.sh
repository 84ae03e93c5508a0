#!/bin/bash
# GPU box: bench + rocprof kernel trace + PMC passes for one workload.  Outputs under gpurun_out/.
#   bash tools/gpu_profile.sh [c2|c3|c4|c5]          then, in the container:  python tools/collect_profiles.py <tag> <workload>
set -o pipefail
WL=${1:-c2}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
EX="--extras none"
python bench.py --workload $WL --steps 20 --warmup 5 $EX 2>&1 | tee gpurun_out/bench_$WL.log &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace_$WL -- python3 bench.py --workload $WL --steps 20 --warmup 5 $EX > gpurun_out/prof_trace_$WL.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch_$WL -- python3 bench.py --workload $WL --inflight 1 --overlap 1 --steps 2 --warmup 1 --no-cpu-baseline --no-alone $EX > gpurun_out/prof_fetch_$WL.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d gpurun_out/prof_write_$WL -- python3 bench.py --workload $WL --inflight 1 --overlap 1 --steps 2 --warmup 1 --no-cpu-baseline --no-alone $EX > gpurun_out/prof_write_$WL.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof_sq_$WL -- python3 bench.py --workload $WL --inflight 1 --overlap 1 --steps 2 --warmup 1 --no-cpu-baseline --no-alone $EX > gpurun_out/prof_sq_$WL.log 2>&1
echo "rc=$?"
