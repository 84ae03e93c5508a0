run() { python bench.py "$@" --steps 16 --warmup 4 --no-alone --no-cpu-baseline --extras none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %.2f ms  %.2f Gsteps/s' % ('$LABEL', d['ms_per_step'], d['value']/1e9))"; }
LABEL="ship three_jobs" run --inflight 3
export LT_HIP_LIBRARY=gpurun_ab/w5/liblt_hip.so
LABEL="w5 three_jobs" run --inflight 3
LABEL="w5 walk_train depth3 bpc4" LT_BENCH_TRAIN_BPC=4 run --inflight 4
LABEL="w5 walk_train depth3 bpc4 nosplit" LT_BENCH_TRAIN_BPC=4 LT_TAIL_SPLIT=0 run --inflight 4
LABEL="w5 walk_train depth2 bpc4 nosplit" LT_BENCH_TRAIN_DEPTH=2 LT_BENCH_TRAIN_BPC=4 LT_TAIL_SPLIT=0 run --inflight 4
LABEL="w5 walk_train depth3 bpc3" LT_BENCH_TRAIN_BPC=3 run --inflight 4
LABEL="w5 walk_train depth3 bpc5" LT_BENCH_TRAIN_BPC=5 run --inflight 4
unset LT_HIP_LIBRARY
LABEL="ship three_jobs" run --inflight 3
