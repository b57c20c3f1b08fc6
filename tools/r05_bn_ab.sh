#!/bin/bash
# A/B of engine switch bn_bwd_fused (BatchNorm-backward sums in the data-gradient epilogue) on the RC-Net and SML steps, alternating
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_bnab; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x -k "bn_bwd or lazy or rounding_oracle or golden or block" > $out/tests.log 2>&1; grep -E "passed|failed|Error" $out/tests.log | tail -3
for i in 1 2; do
 for v in 0 1; do
  timeout 600 python bench.py --gpus 1 --steps 100 --no-sml --no-legs --no-cpu-baseline --opts bn_bwd_fused=$v --full-json $out/full_$v.json 2>$out/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rcnet bn_bwd_fused=$v', d['value'], d['ms_per_step'])"
 done
done
for v in 0 1; do
  timeout 600 python bench.py --gpus 1 --steps 60 --workload sml --no-legs --no-cpu-baseline --opts bn_bwd_fused=$v --full-json $out/full_sml_$v.json 2>>$out/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sml bn_bwd_fused=$v', d['value'], d['ms_per_step'])"
done
python - <<PY
import json
for v in (0,1):
    f=json.load(open("$out/full_%d.json"%v))["roofline"]["families"]
    print(v, " ".join("%s %.3f"%(k,x["ms_per_step"]) for k,x in sorted(f.items(), key=lambda kv:-kv[1]["ms_per_step"])))
PY
