#!/bin/bash
for b in 0 3 5; do echo "== LT_LOG_BITS2=$b"; LT_LOG_BITS2=$b LT_LOG_TIMING=1 timeout -k 10 200 python tools/log_time.py 2>&1 | grep -E "stages|total" ; done
python tools/log_check.py 2>&1 | head -8
