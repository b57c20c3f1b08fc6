#!/bin/bash
# forward-kernel sweep over RC-Net's 3x3 layer shapes: patch-staged kernel (default routing) vs implicit-GEMM (RD_CONV3X3_MIN_BLOCKS huge)
for dt in bf16 fp32; do
for cfg in "240,240,100 32 16" "240,120,50 64 32" "240,120,50 32 64" "240,60,25 128 64" "240,60,25 64 128" "240,30,12 256 128" "240,15,6 384 256" "8,128,256 64 64" "8,64,128 128 128" "8,32,64 128 128"; do
  set -- $cfg
  RD_NHW=$1 RD_CONV3X3_MIN_BLOCKS=0 python3 tools/bench_wgrad.py $2 $3 $dt fwd | sed 's/^/patch /'
  RD_NHW=$1 RD_CONV3X3_MIN_BLOCKS=100000000 python3 tools/bench_wgrad.py $2 $3 $dt fwd | sed 's/^/gemm  /'
done; done
