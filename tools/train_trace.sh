# Kernel timeline of bench.py's walk train (and, for comparison, three jobs in flight):   bash tools/train_trace.sh   [env LT_PART_LDS=1 ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for inf in ${INF:-4 3}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_$inf -- python3 bench.py --inflight $inf --steps 12 --warmup 4 --no-alone --no-cpu-baseline --extras none > gpurun_out/tr_$inf.log 2>&1 || exit 1
  grep '"metric"' gpurun_out/tr_$inf.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'].get('regime'), d['ms_per_step'])"
  python tools/trace_show.py gpurun_out/tr_$inf 36
  rm -rf gpurun_out/tr_$inf
done
