# Which of the LDS partition's resources slows a walk beside it: walk train without tail split, partition variants that drop
# one kind of work (gpurun_ab/dropN: 1 rank atomics, 2 global stores, 3 LDS-DMA loads, 4 sorted copy; results are garbage)
B="--inflight 4 --steps 12 --warmup 4 --no-alone --no-cpu-baseline --extras none"
for v in ${VARIANTS:-ship drop1 drop2 drop3 drop4 drop9}; do
  L=light_transport_amd/liblt_hip.so; [ $v != ship ] && L=gpurun_ab/$v/liblt_hip.so
  LT_HIP_LIBRARY=$L LT_PART_LDS=1 LT_TAIL_SPLIT=0 timeout -k 10 200 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s %.2f ms per job' % ('$v', d['ms_per_step']))" || echo "$v failed"
done
