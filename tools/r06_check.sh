#!/bin/bash
# Round-6 check in one GPU call: (optional) GPU test suite + smoke, the default bench line (compact stdout line + full record; the run collects its own
# kernel trace and counter passes as child processes), optionally a rocprofv3 kernel-stats pass of each stage (--no-children: no profiler inside a profiler).
# usage: tools/r06_check.sh <tag> [tests|sometests|notests] [prof|noprof] [extra bench flags]      (sometests: K="pytest -k expression")
tag=${1:-a}; what=${2:-tests}; prof=${3:-prof}; shift; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/r06_$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ "$what" = "tests" ]; then
  timeout 1800 python -m pytest tests -q -m gpu -x > $out/gpu_tests.log 2>&1
  grep -E "passed|failed" $out/gpu_tests.log | tail -2
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
  tail -1 $out/smoke.log
elif [ "$what" = "sometests" ]; then
  timeout 1200 python -m pytest tests -q -m gpu -x -s -k "${K:-lazy}" > $out/gpu_tests.log 2>&1
  grep -E "passed|failed|Error|convergence|aliased" $out/gpu_tests.log | tail -8
fi
SECONDS=0
timeout 1200 python bench.py --gpus 1 --full-json $out/bench_full.json --detail $out/per_shape.txt --save-kstats $out "$@" > $out/bench.json 2> $out/bench.err
echo "bench.py wall time ${SECONDS}s"
tail -4 $out/bench.err | cut -c1-300; wc -c $out/bench.json; python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().split("\n")[-1])
print("%s: %.1f img/s (%.3f ms)"%(d["metric"],d["value"],d["ms_per_step"]), " ".join("%s %.1f"%(k,d[k]["value"]) for k in ("rcnet","sml","fp32","config4","config4_sml") if k in d))
if d.get("roofline"): print("roofline", d["roofline"]["kernel"], "%.3f"%d["roofline"]["frac"], "avg us %.1f (events %.1f)"%(d["roofline"]["avg_launch_us"], d["roofline"]["avg_launch_us_hip_events"]), "traffic x", d["roofline"].get("traffic_over_algorithmic"))
for f in d.get("roofline_families", []): print("   ", f)
print("unchanged_caller", d.get("unchanged_caller")); print("val_abs_rel", d.get("val_abs_rel")); print("cpu", d.get("cpu_baseline"))
f=json.load(open("$out/bench_full.json"))
for leg in ("rcnet","sml"):
    if leg in f and f[leg].get("families"):
        fam=f[leg]["families"]; print(leg, "launches/step", f[leg].get("launches_per_step"), " ".join("%s %.3f"%(k,v["ms_per_step"]) for k,v in sorted(fam.items(), key=lambda kv:-kv[1]["ms_per_step"])))
PY
if [ "$prof" = "prof" ]; then
  export TMPDIR=/tmp
  for wl in rcnet sml; do
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$wl -o rc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 3 --settle-seconds 0 --no-cpu-baseline --no-legs --no-children --timer-repeat 1 > $out/prof_$wl.log 2>&1
    find $out/prof_$wl -name "*kernel_stats.csv" -exec cp {} $out/${wl}_graphed_kernel_stats.csv \;
    rm -rf $out/prof_$wl
    cd $GRAFT_REPO_ROOT
  done
fi
