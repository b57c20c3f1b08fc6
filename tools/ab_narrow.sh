#!/bin/bash
# narrow-layer (persistent 3x3) kernel A/B on one box: baseline library (RIDERS_HIP_LIB) vs the in-tree build
run() {
  for cfg in "240,240,100 32 16" "240,240,100 16 16" "240,240,100 16 32" "240,240,100 16 1" "240,120,50 64 32" "240,120,50 32 32"; do
    set -- $cfg
    for mode in fwd dgrad wgrad; do
      RD_NHW=$1 python3 tools/bench_wgrad.py $2 $3 bf16 $mode 2>/dev/null | sed "s/^/$LABEL /"
    done
  done
}
LABEL=base RIDERS_HIP_LIB=$PWD/tools/ab/libriders_hip_base.so run
LABEL=new run
