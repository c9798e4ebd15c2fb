#!/bin/bash
# time_variants.sh NAME... : tools/time_mfma.py under the default library, then under
# spt-proto_amd/lib/variants/libspt_hip_NAME.so for each NAME given (explicit names only:
# variant libraries built against an older ABI must never be picked up by a glob)
python tools/time_mfma.py
for name in "$@"; do
  f=spt-proto_amd/lib/variants/libspt_hip_$name.so
  echo "== $f"
  SPT_HIP_LIBRARY=$PWD/$f timeout -k 10 120 python tools/time_mfma.py
done
