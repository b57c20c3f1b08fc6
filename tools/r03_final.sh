#!/bin/bash
# Round-3 evidence in one GPU call: GPU test log, smoke, bench lines (+ per-shape tables), rocprofv3 kernel statistics (RC-Net, SML), per-kernel
# HBM traffic and MFMA-busy counter passes.  Everything lands in gpurun_out/r03_final/ and profiles/r03_*.json; the summaries are then
# copied into profiles/ (tracked) by the caller.
out=$GRAFT_REPO_ROOT/gpurun_out/r03_final; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
timeout 900 python bench.py --detail $out/rcnet_b8_bf16_per_shape.txt 2>/dev/null | grep '"metric"' > $out/bench_rcnet_b8_bf16.json
timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --no-sml --detail $out/rcnet_b8_fp32_per_shape.txt 2>/dev/null | grep '"metric"' > $out/bench_rcnet_b8_fp32.json
timeout 600 python bench.py --dtype fp16 --height 512 --width 1024 --no-cpu-baseline --no-sml 2>/dev/null | grep '"metric"' > $out/bench_rcnet_b8_fp16_512x1024.json
timeout 600 python bench.py --config3 --gpus 1 --no-cpu-baseline --no-sml 2>/dev/null | grep '"metric"' > $out/bench_rcnet_config3_1rank.json
timeout 600 python bench.py --force-ddp --no-cpu-baseline --no-sml 2>/dev/null | grep '"metric"' > $out/bench_rcnet_b8_bf16_rccl_1rank.json
export TMPDIR=/tmp; cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_rc -o rc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --settle-seconds 0 --no-cpu-baseline --no-sml --timer-repeat 1 > $out/p_rc.log 2>&1
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_sml -o sml -- python3 $GRAFT_REPO_ROOT/bench.py --workload sml --steps 10 --warmup 3 --settle-seconds 0 --no-cpu-baseline --timer-repeat 1 > $out/p_sml.log 2>&1
find $out/p_rc -name "*kernel_stats.csv" -exec cp {} $out/rcnet_b8_bf16_kernel_stats.csv \;
find $out/p_sml -name "*kernel_stats.csv" -exec cp {} $out/sml_b16_bf16_kernel_stats.csv \;
rm -rf $out/p_rc $out/p_sml
cd $GRAFT_REPO_ROOT
rm -f profiles/r03_traffic.json profiles/r03_pmc_dominant.json
bash tools/traffic_pass.sh rcnet > $out/traffic.log 2>&1
bash tools/traffic_pass.sh sml >> $out/traffic.log 2>&1
bash tools/pmc_dominant.sh rcnet > $out/pmc.log 2>&1
bash tools/pmc_dominant.sh sml >> $out/pmc.log 2>&1
rm -rf gpurun_out/traffic_rcnet/fetch gpurun_out/traffic_rcnet/write gpurun_out/traffic_sml/fetch gpurun_out/traffic_sml/write
cp profiles/r03_traffic.json profiles/r03_pmc_dominant.json $out/
# the bench line again, now that the counter files exist (roofline.traffic / roofline.mfma_busy filled in)
timeout 900 python bench.py --detail $out/rcnet_b8_bf16_per_shape.txt 2>/dev/null | grep '"metric"' > $out/bench_rcnet_b8_bf16.json
tail -3 $out/gpu_tests.log; tail -1 $out/smoke.log; cut -c1-260 $out/bench_*.json
