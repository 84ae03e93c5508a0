#!/bin/bash
# GPU box: the headline's regimes with two builds of the library, interleaved.   bash tools/ab_bench.sh <variant dir> [steps]
#   (variant = gpurun_ab/<name>, made by tools/build_variant.sh; the shipping build is light_transport_amd/liblt_hip.so)
V=$1; K=${2:-16}
for rep in 1 2; do
  for lib in ship $V; do
    for fl in 3 2; do
      if [ $lib = ship ]; then unset LT_HIP_LIBRARY; else export LT_HIP_LIBRARY=$lib/liblt_hip.so; fi
      python bench.py --inflight $fl --steps $K --warmup 4 --no-alone --no-cpu-baseline --extras none 2>/dev/null |
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib inflight $fl: %.2f ms  %.2f Gsteps/s' % (d['ms_per_step'], d['value']/1e9))"
    done
  done
done
