#!/bin/bash
# lanes-2 tuning sweep on the GPU box: walk workgroups per CU x sub-batches per launch
for bpc in 2 3 4; do for k in 4 8; do
  echo "== LT_OVERLAP_WALK_BPC=$bpc LT_OVERLAP_BATCHES=$k"
  LT_OVERLAP_WALK_BPC=$bpc LT_OVERLAP_BATCHES=$k timeout -k 10 120 python tools/lane_time.py ${1:-c2} 2 2>&1 | grep "lanes 2"
done; done
