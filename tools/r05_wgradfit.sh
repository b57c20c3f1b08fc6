#!/bin/bash
# map-fitted 32x32x16 weight-gradient kernel (option wgrad_fit): tests, conv micro-benchmark (weight gradients only), RC-Net step A/B
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_wgradfit; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x -k "wgrad_fit or wgrad or transpose" > $out/tests.log 2>&1; grep -E "passed|failed|Error" $out/tests.log | tail -3
for rep in 1 2; do
BC_SKIP_CONV=1 timeout 600 python tools/bench_conv.py wgrad_fit=0 wgrad_fit=1 wgrad_fit=0 wgrad_fit=1 2>&1 | grep "wgrad" | cut -c1-110
done
for i in 1 2; do
 for v in 0 1; do
  timeout 600 python bench.py --gpus 1 --steps 100 --no-sml --no-legs --no-cpu-baseline --opts rd.wgrad_fit=$v --full-json $out/full_$v.json 2>$out/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rcnet wgrad_fit=$v', d['value'], d['ms_per_step'])"
 done
done
