#!/bin/bash
# GPU box: walk workgroups per CU x batch layout of one overlapped launch (lt_set_overlap 2).   bash tools/overlap_sweep.sh [c2|c5]
for bpc in 2 3; do for pat in "2,2,1" "1,1" "1,2,2,1" "4,4,1"; do
  echo "== LT_OVERLAP_WALK_BPC=$bpc LT_OVERLAP_PATTERN=$pat"
  LT_OVERLAP_WALK_BPC=$bpc LT_OVERLAP_PATTERN=$pat timeout -k 10 120 python tools/lane_time.py ${1:-c2} 2 2>&1 | grep "lanes 2"
done; done
