#!/bin/bash
# GPU box: tools/lane_time.py / c4_time.py with several builds of the library, interleaved, same box.
#   bash tools/ab_lane.sh "<cases>" <variant dir> [<variant dir> ...]      e.g.  bash tools/ab_lane.sh c2,c5 gpurun_ab/prev
CASES=$1; shift
for rep in 1 2; do
  for lib in ship "$@"; do
    if [ $lib = ship ]; then unset LT_HIP_LIBRARY; else export LT_HIP_LIBRARY=$lib/liblt_hip.so; fi
    echo "== $lib (rep $rep)"
    python tools/lane_time.py $CASES 2
    [ -n "$AB_C4" ] && C4_ROWS=1,2 python tools/c4_time.py
  done
done
