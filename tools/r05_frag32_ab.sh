#!/bin/bash
# A/B of the 32x32x16-MFMA form of the register-fed 3x3 kernel (options frag32_v128 / frag32_v64 = kFrag32Variants index, 0 = the 16x16x32 kernel)
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_frag32; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x -k "frag32" > $out/tests.log 2>&1; grep -E "passed|failed|Error" $out/tests.log | tail -3
for rep in 1 2; do
BC_SKIP_WGRAD=1 timeout 600 python tools/bench_conv.py frag32_v128=0,frag32_v64=0 frag32_v128=1,frag32_v64=3 frag32_v128=2,frag32_v64=3 frag32_v128=0,frag32_v64=0 2>&1 | grep "fwd\|dgrad\|total" | grep -v "64-> 32\|M= 151776\|M=  38192\|M=   9672"
done > $out/bench_conv.log 2>&1
cat $out/bench_conv.log
