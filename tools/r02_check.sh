#!/bin/bash
# Round-2 check in one GPU call: GPU test suite, smoke, the driver's bench command, rocprofv3 kernel statistics.  Output -> gpurun_out/r02_$1/
tag=${1:-a}
out=$GRAFT_REPO_ROOT/gpurun_out/r02_$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x > $out/gpu_tests.log 2>&1
tail -25 $out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
tail -2 $out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail $out/per_shape.txt > $out/bench.json 2> $out/bench.err
tail -3 $out/bench.err; cut -c1-1500 $out/bench.json
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o rc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --settle-seconds 0 --no-cpu-baseline --no-sml > $out/prof.log 2>&1
rm -f $out/prof/*/*kernel_trace.csv $out/prof/*/*agent_info.csv $out/prof/*kernel_trace.csv $out/prof/*agent_info.csv
find $out/prof -name "*kernel_stats.csv" | head -2
