#!/bin/bash
# An A/B build of the library with extra kernel flags, beside the shipping one:  bash tools/build_variant.sh <name> "<flags>"
#   -> gpurun_ab/<name>/liblt_hip.so   (git-ignored, travels to the GPU box; select with LT_HIP_LIBRARY=gpurun_ab/<name>/liblt_hip.so)
set -e
NAME=$1; FLAGS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/light_transport_amd/csrc
OUT=$ROOT/gpurun_ab/$NAME
mkdir -p $OUT
make -s -C $SRC lt_api.o lt_bvh_build.o
KF="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-inline-asm $FLAGS"
( cd $SRC && /opt/rocm/bin/hipcc $KF -c lt_kernels.hip -o $OUT/lt_kernels.o 2>&1 | grep -v "argument unused" || true ) &
( cd $SRC && /opt/rocm/bin/hipcc $KF -c lt_logtally.hip -o $OUT/lt_logtally.o 2>&1 | grep -v "argument unused" || true ) &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liblt_hip.so $OUT/lt_kernels.o $OUT/lt_logtally.o $SRC/lt_api.o $SRC/lt_bvh_build.o -ldl 2>&1 | grep -v "argument unused" || true
ls -la $OUT/liblt_hip.so
