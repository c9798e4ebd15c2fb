#!/bin/bash
# tools/variant.sh NAME UNIT FLAGS...: libspt_hip.so with translation unit UNIT (e.g. grouped_gemm)
# recompiled under extra FLAGS -> spt-proto_amd/lib/exp/libspt_hip_NAME.so (travels to the GPU
# box; use with SPT_HIP_LIBRARY=...).  For A/B timing of one kernel; nothing ships from here.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; unit=$2; shift 2
mkdir -p $R/spt-proto_amd/lib/exp
cd $R/spt-proto_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-fast-math \
    -fno-slp-vectorize -Wall -Wno-unused-function "$@" -c $unit.hip -o ../lib/exp/${unit}_$name.o
objs=$(ls ../lib/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp/libspt_hip_$name.so $objs ../lib/exp/${unit}_$name.o
rm ../lib/exp/${unit}_$name.o
echo built lib/exp/libspt_hip_$name.so
