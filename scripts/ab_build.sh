#!/bin/bash
# A/B builds of one kernel source for same-box comparisons (box-to-box spread on this pool is 3-5 %, larger than most kernel
# changes): scripts/ab_build.sh NAME SRC.hip [-DFLAG=...] -> build_ab/NAME.so (all other objects from the normal build).
# Run both with MLA_HIP_LIB=build_ab/NAME.so inside ONE gpurun call.
set -e
cd "$(dirname "$0")/../multimodal-learning-with-alternating-unimodal-adaptation_amd/csrc"
NAME=$1; SRC=$2; shift 2
make -s >/dev/null
mkdir -p ../../build_ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $SRC -o ../../build_ab/$NAME.o
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../build_ab/$NAME.o -o ../../build_ab/$NAME.so 2>/dev/null
rm ../../build_ab/$NAME.o
echo built build_ab/$NAME.so
