#!/bin/bash
# builds liblt_hip.so variants with different occupancy targets into light_transport_amd/variants/
set -e
cd "$(dirname "$0")/../light_transport_amd/csrc"
mkdir -p ../variants
for v in "3 4" "4 4" "4 5" "3 5"; do
  set -- $v
  rm -f lt_kernels.o
  make -s EXTRA_KFLAGS="-DLT_F64_WAVES=$1 -DLT_F32_WAVES=$2"
  cp ../liblt_hip.so ../variants/liblt_hip_f64w$1_f32w$2.so
done
rm -f lt_kernels.o; make -s
