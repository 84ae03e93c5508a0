#!/bin/bash
# GPU box: bench.py plain against bench.py under its own launcher at world size 1 (torch.distributed.run + RCCL, every job's grid
# reduced on its stream), interleaved, both workloads, each regime pinned.   bash tools/world1_ab.sh > profiles/rNN_world1_ab.log
one() {  # $1 = plain | launched, rest = bench flags
  kind=$1; shift
  if [ $kind = plain ]; then out=$(python bench.py --gpus 1 "$@" 2>/dev/null)
  else out=$(python -c "import sys, bench; sys.exit(bench.self_launch(1, sys.argv[1:]))" --gpus 1 "$@" 2>/dev/null); fi
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('%-9s %-28s %7.2f ms  %6.2f Gsteps/s  overlap factor %s  reduces %s (%s)' % ('$kind', ' '.join(sys.argv[1:]), d['ms_per_step'], d['value']/1e9, c['overlap_factor_per_rank'], c['reduce']['calls_rank0'], c['reduce']['backend']))" "$@"
}
for rep in 1 2; do
  for fl in "--inflight 3" "--inflight 2" "--inflight 4" "--inflight 1"; do
    for kind in plain launched; do one $kind $fl --steps 16 --warmup 4 --no-alone --no-cpu-baseline --extras none; done
  done
done
for fl in "--inflight 2" "--inflight 1"; do
  for kind in plain launched; do one $kind --workload c5 $fl --steps 8 --warmup 3 --no-alone --no-cpu-baseline --extras none; done
done
