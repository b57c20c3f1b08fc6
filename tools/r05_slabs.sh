#!/bin/bash
# split-K study: the 8 x TW weight-gradient kernel with fewer persistent blocks per slice (option conv3x3_g8 = blocks per XCD and slice): slab traffic vs parallelism
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_slabs; mkdir -p $out
for rep in 1 2; do
BC_SKIP_CONV=1 timeout 600 python tools/bench_conv.py default conv3x3_g8=2 conv3x3_g8=1 conv3x3_g8=8 conv3x3_g8= 2>&1 | grep "wgrad M=  21600\|wgrad M=  86400\|wgrad M= 360000" | cut -c1-110
done
