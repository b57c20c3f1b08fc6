#!/bin/bash
# per-shape comparison of the register-fed 3x3 kernel's block shapes on the > 64-channel decoder layers (block-count quantisation study)
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_fragvar; mkdir -p $out
for rep in 1 2; do
BC_SKIP_WGRAD=1 timeout 900 python tools/bench_conv.py default frag_split_blocks=100000 frag_split_blocks=,frag32_v128=2 frag32_v128=,frag_v128=0 frag_v128=1 2>&1 | grep "fwd\|dgrad\|total" | grep "M=  21600\|M=  86400\|total"
done > $out/bench_conv.log 2>&1
cat $out/bench_conv.log | cut -c1-120
