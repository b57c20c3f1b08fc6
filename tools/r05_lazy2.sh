#!/bin/bash
# re-measure of the consumer-side BatchNorm apply on every conv -> BatchNorm -> conv chain (engine switch lazy_bn = 2) against the default (1), RC-Net step, alternating
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_lazy2; mkdir -p $out
for i in 1 2; do
 for v in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 100 --no-sml --no-legs --no-cpu-baseline --opts lazy_bn=$v --full-json $out/full_$v.json 2>$out/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rcnet lazy_bn=$v', d['value'], d['ms_per_step'])"
 done
done
