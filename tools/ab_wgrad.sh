#!/bin/bash
# A/B of several builds of the library on the weight-gradient shapes of tools/bench_conv.py: tools/ab/lib_<name>.so ... vs the current build.
# usage: tools/ab_wgrad.sh name1 name2 ...   (the current build is always measured last as "new")
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for n in "$@"; do
    RIDERS_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_$n.so BC_SKIP_CONV=1 python tools/bench_conv.py RD_X=$n 2>&1 | grep "wgrad"
  done
  BC_SKIP_CONV=1 python tools/bench_conv.py RD_X=new 2>&1 | grep "wgrad"
done
