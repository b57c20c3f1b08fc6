#!/bin/bash
# Round-end evidence in one GPU call: test log, bench lines (+ per-shape tables), rocprofv3 kernel statistics, PMC traffic of the
# dominant launch shape.  Everything lands in gpurun_out/final/; the summaries are then copied into profiles/ (tracked).
out=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
timeout 600 python bench.py --detail $out/rcnet_b8_bf16_per_shape.txt 2>/dev/null | grep '"metric"' > $out/bench_rcnet_b8_bf16.json
timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --detail $out/rcnet_b8_fp32_per_shape.txt 2>/dev/null | grep '"metric"' > $out/bench_rcnet_b8_fp32.json
timeout 600 python bench.py --workload sml --detail $out/sml_b16_bf16_per_shape.txt 2>/dev/null | grep '"metric"' > $out/bench_sml_b16_bf16.json
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_rc_bf16 -o rc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 15 --warmup 3 --no-cpu-baseline > $out/p_rc_bf16.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_rc_fp32 -o rc -- python3 $GRAFT_REPO_ROOT/bench.py --dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline > $out/p_rc_fp32.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_sml -o sml -- python3 $GRAFT_REPO_ROOT/bench.py --workload sml --steps 10 --warmup 3 --no-cpu-baseline > $out/p_sml.log 2>&1
rm -f $out/p_*/*kernel_trace.csv $out/p_*/*agent_info.csv
cd $GRAFT_REPO_ROOT
timeout 500 bash tools/pmc_small.sh 240,15,6 384 256 bf16 wgrad gpurun_out/final/pmc_wgrad384 > $out/pmc.log 2>&1
timeout 300 bash tools/pmc_small.sh 240,240,100 32 16 bf16 dgrad gpurun_out/final/pmc_dgrad32 >> $out/pmc.log 2>&1
tail -3 $out/gpu_tests.log; cat $out/smoke.log | tail -1; cut -c1-200 $out/bench_*.json
