#!/bin/bash
# build_variant.sh NAME FILE.hip "-DFLAG ..." : libspt_hip with one translation unit rebuilt
# under extra flags -> spt-proto_amd/lib/variants/libspt_hip_NAME.so (use with SPT_HIP_LIBRARY)
set -e
cd "$(dirname "$0")/../spt-proto_amd/csrc"
make -s >/dev/null
mkdir -p ../lib/variants
base=$(basename "$2" .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-fast-math -fno-slp-vectorize $3 -c "$2" -o ../lib/variants/${base}_$1.o 2>/dev/null
objs=$(ls ../lib/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libspt_hip_$1.so $objs ../lib/variants/${base}_$1.o 2>/dev/null
echo built $1
