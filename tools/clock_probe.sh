# Core clock and package power while bench.py runs a regime (is the job bound by the power cap?):  bash tools/clock_probe.sh
mkdir -p gpurun_out
probe() {   # $1 label, rest: bench args
  label=$1; shift
  ( for i in $(seq 1 400); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket Power|mclk|fclk" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clk_$label.log &
  SP=$!
  timeout -k 10 300 python bench.py "$@" --steps 200 --warmup 4 --no-alone --no-cpu-baseline --extras none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['ms_per_step'], d['config'].get('regime'))"
  kill $SP 2>/dev/null; wait $SP 2>/dev/null
  echo "--- $label: samples (head / middle / tail)"; sed -n '1p' gpurun_out/clk_$label.log; awk 'NR%6==0' gpurun_out/clk_$label.log | tail -12
}
probe idle_then_three --inflight 3
probe walk_only --inflight 1 --overlap 1 --tally-mode atomic
