# "Full train": walks one after another at FOUR workgroups per CU (112-VGPR slab walk, variant q112), the previous job's partition and
# reduce as one 64-VGPR wave per SIMD beside them (part_lds bits 0 + 2).  A/B against the shipping regimes.
B="--steps 16 --warmup 4 --no-alone --no-cpu-baseline --extras none"
run() { env $ENVS timeout -k 10 300 python bench.py --workload ${W:-c2} --inflight $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-64s %.2f ms  %.2f Gsteps/s  %s' % ('$LABEL', d['ms_per_step'], d['value']/1e9, d['config'].get('regime')))" || echo "$LABEL failed"; }
V=gpurun_ab/q112/liblt_hip.so
LABEL="ship three_jobs" ENVS="A=1" run 3
LABEL="q112 three_jobs" ENVS="LT_HIP_LIBRARY=$V" run 3
LABEL="q112 one_at_a_time (walk 4/CU alone)" ENVS="LT_HIP_LIBRARY=$V" run 1 
LABEL="q112 full train bpc4 depth3 nosplit part_lds=5" ENVS="LT_HIP_LIBRARY=$V LT_BENCH_TRAIN_BPC=4 LT_TAIL_SPLIT=0 LT_PART_LDS=5" run 4
LABEL="q112 full train bpc4 depth2 nosplit part_lds=5" ENVS="LT_HIP_LIBRARY=$V LT_BENCH_TRAIN_BPC=4 LT_BENCH_TRAIN_DEPTH=2 LT_TAIL_SPLIT=0 LT_PART_LDS=5" run 4
LABEL="q112 full train bpc4 depth3 split part_lds=5" ENVS="LT_HIP_LIBRARY=$V LT_BENCH_TRAIN_BPC=4 LT_PART_LDS=5" run 4
LABEL="ship train bpc4 nosplit part_lds=5 (128-VGPR walk)" ENVS="LT_BENCH_TRAIN_BPC=4 LT_TAIL_SPLIT=0 LT_PART_LDS=5" run 4
