#!/bin/bash
# time_block_variants.sh NAME... : sparse TransformerBlock step + routed FFN under the default
# library and under spt-proto_amd/lib/exp/libspt_hip_NAME.so
export TUNINGS=sparse
python tools/bench_block.py | python -c "import json,sys; d=json.load(sys.stdin); print('default block ms', d['sparse']['ms_per_step'])"
python tools/bench_ffn.py | tail -1 | cut -c1-400
for name in "$@"; do
  export SPT_HIP_LIBRARY=$PWD/spt-proto_amd/lib/exp/libspt_hip_$name.so
  python tools/bench_block.py | python -c "import json,sys; d=json.load(sys.stdin); print('$name block ms', d['sparse']['ms_per_step'])"
  python tools/bench_ffn.py | tail -1 | cut -c1-400
done
