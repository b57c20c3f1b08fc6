#!/bin/bash
# fused decoder head (rd_head.hip): parity subset, kernel micro-benchmark, A/B of engine switch bn_head on the RC-Net step (alternating)
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_head; mkdir -p $out
timeout 1200 python -m pytest tests -q -m gpu -x -k "bn_head or lazy or rounding_oracle or golden or block or rcnet" > $out/tests.log 2>&1; grep -E "passed|failed|Error" $out/tests.log | tail -3
timeout 600 python tools/bench_head.py > $out/bench_head.txt 2>&1; cat $out/bench_head.txt
for i in 1 2; do
 for v in 0 1; do
  timeout 600 python bench.py --gpus 1 --steps 100 --no-sml --no-legs --no-cpu-baseline --opts bn_head=$v --full-json $out/full_$v.json 2>$out/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rcnet bn_head=$v', d['value'], d['ms_per_step'])"
 done
done
python - <<PY
import json
for v in (0,1):
    f=json.load(open("$out/full_%d.json"%v))["roofline"]["families"]
    print(v, " ".join("%s %.3f"%(k,x["ms_per_step"]) for k,x in sorted(f.items(), key=lambda kv:-kv[1]["ms_per_step"])))
PY
