#!/bin/bash
# generic bf16 weight-gradient kernel (small maps, stride 2, 7x7) A/B on one box: baseline library vs the in-tree build
run() {
  for cfg in "240,15,6 384 256" "240,15,6 256 256" "8,16,32 128 128" "8,8,16 256 256"; do
    set -- $cfg
    RD_NHW=$1 python3 tools/bench_wgrad.py $2 $3 bf16 wgrad 2>/dev/null | sed "s/^/$LABEL wgrad /" | sed 's/wgrad wgrad/wgrad/'
  done
}
LABEL=base RIDERS_HIP_LIB=$PWD/tools/ab/libriders_hip_base.so run
LABEL=new run
