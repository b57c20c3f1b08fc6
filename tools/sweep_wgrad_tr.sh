#!/bin/bash
# bf16 weight gradient of RC-Net's narrow 3x3 layers (transpose-read kernel) + persistent-grid size experiment for the forward kernel
for cfg in "240,240,100 16 16" "240,240,100 32 16" "240,240,100 16 1" "240,120,50 64 32" "240,120,50 32 32"; do
  set -- $cfg
  RD_NHW=$1 python3 tools/bench_wgrad.py $2 $3 bf16 wgrad 2>/dev/null
done
for g8 in 64 128 192 256 384; do
  echo "G8=$g8"
  RD_CONV3X3_G8=$g8 RD_NHW=240,240,100 python3 tools/bench_wgrad.py 16 16 bf16 fwd 2>/dev/null
  RD_CONV3X3_G8=$g8 RD_NHW=240,240,100 python3 tools/bench_wgrad.py 32 16 bf16 fwd 2>/dev/null
  RD_CONV3X3_G8=$g8 RD_NHW=240,240,100 python3 tools/bench_wgrad.py 16 16 bf16 wgrad 2>/dev/null
done
