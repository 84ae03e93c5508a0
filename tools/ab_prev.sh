# same-box A/B: a previous library (gpurun_ab/prev4) against the working tree's, every workload, three jobs in flight + one call
B="--steps 20 --warmup 5 --no-alone --no-cpu-baseline --extras none"
run() { env $ENVS timeout -k 10 300 python bench.py --workload $W --inflight $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %-3s %.2f ms  %.2f Gsteps/s  %s' % ('$LABEL', '$W', d['ms_per_step'], d['value']/1e9, d['config'].get('regime')))" || echo "$LABEL $W failed"; }
for W in ${WL:-c2 c3 c4}; do
  LABEL="prev three_jobs" ENVS="LT_HIP_LIBRARY=gpurun_ab/${PREV:-prev4}/liblt_hip.so" run 3
  LABEL="new  three_jobs" ENVS="A=1" run 3
  LABEL="prev one_call" ENVS="LT_HIP_LIBRARY=gpurun_ab/${PREV:-prev4}/liblt_hip.so" run 1
  LABEL="new  one_call" ENVS="A=1" run 1
done
W=c5
LABEL="prev two_jobs" ENVS="LT_HIP_LIBRARY=gpurun_ab/${PREV:-prev4}/liblt_hip.so" run 2
LABEL="new  two_jobs" ENVS="A=1" run 2
W=c2
LABEL="prev three_jobs" ENVS="LT_HIP_LIBRARY=gpurun_ab/${PREV:-prev4}/liblt_hip.so" run 3
LABEL="new  three_jobs" ENVS="A=1" run 3
