#!/bin/bash
# A/B of the register-fed 3x3 kernel's double-buffered patch (option frag_db) on the conv micro-benchmark and the RC-Net step
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_fragdb; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x -k "frag or lazy or conv" > $out/tests.log 2>&1; grep -E "passed|failed|Error" $out/tests.log | tail -3
for rep in 1 2; do
BC_SKIP_WGRAD=1 timeout 600 python tools/bench_conv.py frag_db=1 frag_db=0 frag_db=1 frag_db=0 2>&1 | grep "total"
done
BC_SKIP_WGRAD=1 timeout 600 python tools/bench_conv.py frag_db=0 frag_db=1 2>&1 | grep "fwd\|dgrad" | cut -c1-100
for i in 1 2; do
 for v in 0 1; do
  timeout 600 python bench.py --gpus 1 --steps 100 --no-sml --no-legs --no-cpu-baseline --opts rd.frag_db=$v --full-json $out/full_$v.json 2>$out/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rcnet frag_db=$v', d['value'], d['ms_per_step'], d['roofline_conv']['frac'])"
 done
done
