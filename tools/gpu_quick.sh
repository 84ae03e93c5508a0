#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/gpu_tests_tail.log &&
python tools/sweep.py tally 2>&1 | tee gpurun_out/sweep_tally.log &&
LT_DIAG_NO_TALLY=1 python tools/sweep.py tally 2>&1 | tee gpurun_out/sweep_notally.log &&
python tools/sweep.py configs 2>&1 | tee gpurun_out/sweep_configs.log
