#!/bin/bash
# A/B of two builds of the library on the conv micro-benchmark: tools/ab/lib_base.so (copied before a change) vs the current build.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  RIDERS_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_base.so BC_SKIP_WGRAD=1 python tools/bench_conv.py RD_X=base 2>&1 | grep -v wgrad | grep "total\|fwd\|dgrad"
  BC_SKIP_WGRAD=1 python tools/bench_conv.py RD_X=new 2>&1 | grep -v wgrad | grep "total\|fwd\|dgrad"
done
