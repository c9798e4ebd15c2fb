#!/bin/bash
# time_mfma_ab.sh NAME : tools/time_mfma.py alternating between the default library and
# spt-proto_amd/lib/exp/libspt_hip_NAME.so on the same box (A B A B)
for i in 1 2; do
  python tools/time_mfma.py 2>/dev/null | tail -1
  SPT_HIP_LIBRARY=$PWD/spt-proto_amd/lib/exp/libspt_hip_$1.so python tools/time_mfma.py 2>/dev/null | tail -1
done
