#!/bin/bash
# exact-2x up-convolutions on their source (engine switches up2_on_source = forward, up2_dgrad = data gradient): GPU suite, then A/B on the RC-Net step (alternating)
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_up2; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu > $out/tests.log 2>&1; grep -E "passed|failed|Error" $out/tests.log | tail -5
for i in 1 2; do
 for v in "0,0" "1,0" "1,1"; do
  f=${v%,*}; d=${v#*,}
  timeout 600 python bench.py --gpus 1 --steps 100 --no-sml --no-legs --no-cpu-baseline --no-live-traffic --opts up2_on_source=$f,up2_dgrad=$d --full-json $out/full_$f$d.json 2>$out/err_$f$d.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rcnet up2_on_source=$f up2_dgrad=$d', d['value'], d['ms_per_step'])"
 done
done
python - <<PY
import json
for v in ("00","10","11"):
    d=json.load(open("$out/full_%s.json"%v))
    f=d["roofline"]["families"]
    print(v, " ".join("%s %.3f"%(k,x["ms_per_step"]) for k,x in sorted(f.items(), key=lambda kv:-kv[1]["ms_per_step"])))
PY
