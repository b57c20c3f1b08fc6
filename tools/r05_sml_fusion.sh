#!/bin/bash
# SML FeatureFusionBlock: the 1x1 out_conv before the bilinear x2 (engine switch fusion_conv_first): SML GPU tests, then A/B on the SML step (alternating)
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_sml_fusion; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_sml.py -q -m gpu > $out/tests.log 2>&1; grep -E "passed|failed|Error" $out/tests.log | tail -3
for i in 1 2; do
 for v in 0 1; do
  timeout 600 python bench.py --gpus 1 --steps 60 --workload sml --no-legs --no-cpu-baseline --no-live-traffic --opts fusion_conv_first=$v 2>$out/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sml fusion_conv_first=$v', d['value'], d['ms_per_step'])"
 done
done
