#!/bin/bash
# forward of the exact-2x up-convolutions on their source (engine switch up2_on_source): GPU suite, then A/B on the RC-Net step (alternating)
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_up2; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu > $out/tests.log 2>&1; grep -E "passed|failed|Error" $out/tests.log | tail -5
for i in 1 2; do
 for v in 0 1; do
  timeout 600 python bench.py --gpus 1 --steps 100 --no-sml --no-legs --no-cpu-baseline --no-live-traffic --opts up2_on_source=$v --full-json $out/full_$v.json 2>$out/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rcnet up2_on_source=$v', d['value'], d['ms_per_step'])"
 done
done
python - <<PY
import json
for v in (0,1):
    d=json.load(open("$out/full_%d.json"%v))
    sh=[(s["shape"],s["avg_us"]) for s in d["roofline"].get("shapes",[])]
    f=d["roofline"]["families"]
    print(v, " ".join("%s %.3f"%(k,x["ms_per_step"]) for k,x in sorted(f.items(), key=lambda kv:-kv[1]["ms_per_step"])))
PY
grep -h "on source\|Cin=32 Cout=16 k=3 s=1 up" gpurun_out/bench_detail* 2>/dev/null | head
