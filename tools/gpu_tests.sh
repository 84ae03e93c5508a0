#!/bin/bash
# run on the GPU box: gpurun -- 'bash tools/gpu_tests.sh'
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tee gpurun_out/gpu_tests.log
