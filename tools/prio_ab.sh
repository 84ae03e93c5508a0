# A/B of raised wave priority in the walk kernels (variant build -DLT_WALK_PRIO=3) in the jobs-in-flight regimes
B="--steps 16 --warmup 4 --no-alone --no-cpu-baseline --extras none"
run() { env $ENVS timeout -k 10 300 python bench.py --workload ${W:-c2} --inflight $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %.2f ms  %.2f Gsteps/s  %s' % ('$LABEL', d['ms_per_step'], d['value']/1e9, d['config'].get('regime')))" || exit 1; }
V=gpurun_ab/${VARIANT:-prio3}/liblt_hip.so
LABEL="ship three_jobs" ENVS="A=1" run 3
LABEL="prio three_jobs" ENVS="LT_HIP_LIBRARY=$V" run 3
LABEL="prio three_jobs part_lds" ENVS="LT_HIP_LIBRARY=$V LT_PART_LDS=1" run 3
LABEL="ship walk_train" ENVS="A=1" run 4
LABEL="prio walk_train" ENVS="LT_HIP_LIBRARY=$V" run 4
LABEL="prio walk_train part_lds" ENVS="LT_HIP_LIBRARY=$V LT_PART_LDS=1" run 4


LABEL="ship one_call" ENVS="A=1" run 1
LABEL="ship three_jobs" ENVS="A=1" run 3
