#!/bin/bash
# block-count quantisation of the register-fed 3x3 kernel: the 30x12 decoder layers (256 -> 128, 128 x 128 blocks, 2 per CU = 512 slots) at RoI counts
# that give 510 / 756 / 1023 / 1133 tiles (162 / 240 / 325 / 360 RoIs); TFLOP/s per shape
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_quant; mkdir -p $out
for rep in 1 2; do
for n in 162 240 325 360 240 162; do
  BC_ROIS=$n BC_SKIP_WGRAD=1 timeout 300 python tools/bench_conv.py rois=$n 2>&1 | grep "M= *[0-9]* 256->128\|M= *[0-9]* 128->256" | sed "s/^/N=$n /"
done
done > $out/quant.log 2>&1
cat $out/quant.log | cut -c1-120
