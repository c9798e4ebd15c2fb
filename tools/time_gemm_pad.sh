#!/bin/bash
# leading-dimension sweep of the k-contiguous grouped GEMM (channel-camping check)
for pad in 0 16 32 64 96; do PAD=$pad python tools/bench_gemm.py; done
