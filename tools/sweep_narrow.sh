#!/bin/bash
# RC-Net decoder-tail layers (ROI resolution): narrow-layer kernel vs implicit GEMM
for dt in bf16 fp32; do
for cfg in "240,240,100 32 16" "240,240,100 16 16" "240,240,100 16 1" "240,240,100 16 32" "240,240,100 1 16" "240,120,50 64 32"; do
  set -- $cfg
  RD_NHW=$1 python3 tools/bench_wgrad.py $2 $3 $dt fwd 2>/dev/null | sed 's/^/new   /'
  RD_NHW=$1 RD_CONV3X3_MIN_BLOCKS=100000000 python3 tools/bench_wgrad.py $2 $3 $dt fwd 2>/dev/null | sed 's/^/gemm  /'
done; done
