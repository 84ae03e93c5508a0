B="--steps 20 --warmup 5 --no-alone --no-cpu-baseline --extras none"
run() { env $ENVS timeout -k 10 300 python bench.py --workload c2 --inflight $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %.2f ms  %.2f Gsteps/s  %s' % ('$LABEL', d['ms_per_step'], d['value']/1e9, d['config'].get('regime')))" || echo "$LABEL failed"; }
LABEL="three_jobs" ENVS="A=1" run 3
LABEL="three_jobs part_lds=1" ENVS="LT_PART_LDS=1" run 3
LABEL="two_jobs" ENVS="A=1" run 2
LABEL="walk_train" ENVS="A=1" run 4
LABEL="walk_train part_lds=1" ENVS="LT_PART_LDS=1" run 4
LABEL="three_jobs tail_split=0" ENVS="LT_TAIL_SPLIT=0" run 3
LABEL="three_jobs" ENVS="A=1" run 3
