#!/bin/bash
# bf16 weight gradient of RC-Net's wide 3x3 layers: sliced transpose-read kernel (default) -- compare with profiles/ per-shape tables
for cfg in "240,60,25 128 64" "240,30,12 256 128" "240,120,50 64 32" "8,64,128 64 64" "8,32,64 128 128" "8,16,32 128 128" "240,15,6 384 256"; do
  set -- $cfg
  RD_NHW=$1 python3 tools/bench_wgrad.py $2 $3 bf16 wgrad 2>/dev/null
done
