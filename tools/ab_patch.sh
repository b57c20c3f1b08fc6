#!/bin/bash
# patch-staged 3x3 kernel A/B on one box: baseline library (RIDERS_HIP_LIB) vs the in-tree build, forward + data gradient
run() {
  for cfg in "240,120,50 64 32" "240,120,50 32 64" "240,60,25 128 64" "240,60,25 64 128" "240,30,12 256 128" "240,15,6 384 256" "8,128,256 64 64" "8,64,128 128 128" "8,32,64 128 128"; do
    set -- $cfg
    for mode in fwd dgrad; do
      RD_NHW=$1 python3 tools/bench_wgrad.py $2 $3 bf16 $mode 2>/dev/null | sed "s/^/$LABEL /"
    done
  done
}
LABEL=base RIDERS_HIP_LIB=$PWD/tools/ab/libriders_hip_base.so run
LABEL=new run
