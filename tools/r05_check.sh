#!/bin/bash
# Round-5 check in one GPU call: (optional) GPU test suite + smoke, the default bench line (compact stdout line + full record), optional rocprofv3 kernel stats.
# usage: tools/r05_check.sh <tag> [tests|sometests|notests] [prof|noprof] [extra bench flags]      (sometests: K="pytest -k expression")
tag=${1:-a}; what=${2:-tests}; prof=${3:-prof}; shift; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/r05_$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ "$what" = "tests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -x > $out/gpu_tests.log 2>&1
  grep -E "passed|failed" $out/gpu_tests.log | tail -2
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
  tail -1 $out/smoke.log
elif [ "$what" = "sometests" ]; then
  timeout 900 python -m pytest tests -q -m gpu -x -k "${K:-lazy}" > $out/gpu_tests.log 2>&1
  grep -E "passed|failed|Error" $out/gpu_tests.log | tail -4
fi
timeout 900 python bench.py --gpus 1 --full-json $out/bench_full.json --detail $out/per_shape.txt "$@" > $out/bench.json 2> $out/bench.err
tail -3 $out/bench.err | cut -c1-300; wc -c $out/bench.json; python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().split("\n")[-1])
print("rcnet %.1f img/s (%.3f ms)"%(d["value"],d["ms_per_step"]), " ".join("%s %.1f"%(k,d[k]["value"]) for k in ("sml","chained","fp32","config4","config4_sml") if k in d))
print("roofline", d["roofline"]["kernel"], "%.3f"%d["roofline"]["frac"], "| conv", d["roofline_conv"]["kernel"], "%.3f"%d["roofline_conv"]["frac"])
f=json.load(open("$out/bench_full.json"))
fam=f["roofline"]["families"]
print(" ".join("%s %.3f"%(k,v["ms_per_step"]) for k,v in sorted(fam.items(), key=lambda kv:-kv[1]["ms_per_step"])))
PY
if [ "$prof" = "prof" ]; then
  export TMPDIR=/tmp; cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o rc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --settle-seconds 0 --no-cpu-baseline --no-sml --no-legs --timer-repeat 1 "$@" > $out/prof.log 2>&1
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
  rm -rf $out/prof
fi
