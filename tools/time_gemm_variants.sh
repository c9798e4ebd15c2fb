#!/bin/bash
# time_gemm_variants.sh NAME... : tools/bench_gemm.py under the default library, then under
# spt-proto_amd/lib/exp/libspt_hip_NAME.so for each NAME (timing experiments: -DGG_EXP_* builds)
python tools/bench_gemm.py
for name in "$@"; do
  SPT_HIP_LIBRARY=$PWD/spt-proto_amd/lib/exp/libspt_hip_$name.so timeout -k 10 120 python tools/bench_gemm.py
done
