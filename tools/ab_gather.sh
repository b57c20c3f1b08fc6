#!/bin/bash
# A/B of two library builds on one box: RIDERS_HIP_LIB=tools/ab/libriders_hip_base.so (baseline) vs the in-tree build
run() {  # label, env...
  for cfg in "240,240,100 32 16" "240,240,100 16 16" "240,240,100 16 32" "240,120,50 64 32" "240,60,25 128 64" "240,30,12 256 128" "8,128,256 64 64" "8,64,128 128 128"; do
    set -- $cfg
    for mode in fwd dgrad wgrad; do
      RD_NHW=$1 python3 tools/bench_wgrad.py $2 $3 bf16 $mode 2>/dev/null | sed "s/^/$LABEL $mode /"
    done
  done
}
LABEL=base RIDERS_HIP_LIB=$PWD/tools/ab/libriders_hip_base.so run
LABEL=new run
for i in 1 2; do
  RIDERS_HIP_LIB=$PWD/tools/ab/libriders_hip_base.so python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | sed 's/^/base /' | cut -c1-200
  python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | sed 's/^/new  /' | cut -c1-200
done
RIDERS_HIP_LIB=$PWD/tools/ab/libriders_hip_base.so python3 bench.py --workload sml --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | sed 's/^/base /' | cut -c1-200
python3 bench.py --workload sml --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | sed 's/^/new  /' | cut -c1-200
