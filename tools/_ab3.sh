set -e
L=gpurun_out/w5.log
echo "== 96 VGPR walk, LT_OVERLAP_WALK_BPC=2" > $L
LT_OVERLAP_WALK_BPC=2 timeout -k 10 300 python tools/lane_time.py c2,c5 3 >> $L 2>&1
echo "== 96 VGPR walk, LT_OVERLAP_WALK_BPC=2 LT_OVERLAP_BATCHES=3" >> $L
LT_OVERLAP_BATCHES=3 LT_OVERLAP_WALK_BPC=2 timeout -k 10 300 python tools/lane_time.py c2 3 >> $L 2>&1
echo "== 96 VGPR walk, LT_OVERLAP_WALK_BPC=2 LT_OVERLAP_BATCHES=6" >> $L
LT_OVERLAP_BATCHES=6 LT_OVERLAP_WALK_BPC=2 timeout -k 10 300 python tools/lane_time.py c2 3 >> $L 2>&1
cat $L
