# A/B of the LDS-DMA staged partition (lt_set_tuning "part_lds" / LT_PART_LDS=1): stage times of one launch, then the
# jobs-in-flight regimes of bench.py.     bash tools/part_lds_ab.sh [workloads, default "c2"]
mkdir -p gpurun_out
timeout -k 10 300 python tools/part_lds_check.py c2 || exit 1
B="--steps 16 --warmup 4 --no-alone --no-cpu-baseline --extras none"
for w in ${1:-c2}; do
for cfg in "3:" "4:" "4:LT_PART_LDS=1" "3:LT_PART_LDS=1" "2:LT_PART_LDS=1"; do
  inf=${cfg%%:*}; envs=${cfg#*:}
  echo "== $w inflight $inf env [$envs]"
  env $envs timeout -k 10 300 python bench.py --workload $w --inflight $inf $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value']/1e9, d['ms_per_step'], d['config'].get('regime'))" || exit 1
done
done
