#!/bin/bash
# Round-5 evidence in ONE GPU call (run last, on the final kernel sources: the counter files carry a sha1 of riders_amd/csrc):
#   GPU suite + smoke, rocprofv3 kernel statistics of the bf16 RC-Net and SML legs, FETCH_SIZE / WRITE_SIZE passes (per-kernel HBM traffic),
#   the MFMA-busy / wait counter pass, then the default bench line (which attaches the fresh counter files to its roofline objects).
# Everything lands under gpurun_out/r05_final/ ; copy into profiles/ afterwards (see the end of this script for the names).
out=$GRAFT_REPO_ROOT/gpurun_out/r05_final; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
grep -E "passed|failed" $out/gpu_tests.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
export TMPDIR=/tmp
for wl in rcnet sml; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$wl -o rc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 3 --settle-seconds 0 --no-cpu-baseline --no-sml --no-legs --timer-repeat 1 > $out/prof_$wl.log 2>&1
  find $out/prof_$wl -name "*kernel_stats.csv" -exec cp {} $out/${wl}_kernel_stats.csv \;
  rm -rf $out/prof_$wl
  cd $GRAFT_REPO_ROOT
  bash tools/traffic_pass.sh $wl > $out/traffic_$wl.log 2>&1; tail -1 $out/traffic_$wl.log | cut -c1-300
  bash tools/pmc_dominant.sh $wl > $out/pmc_$wl.log 2>&1; tail -1 $out/pmc_$wl.log | cut -c1-300
  rm -rf gpurun_out/traffic_$wl/fetch gpurun_out/traffic_$wl/write
done
cp profiles/r05_traffic.json profiles/r05_pmc_dominant.json $out/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --detail $out/per_shape.txt --full-json $out/bench_full.json > $out/bench.json 2> $out/bench.err
tail -2 $out/bench.err | cut -c1-200; cut -c1-400 $out/bench.json
timeout 600 python bench.py --gpus 1 --steps 50 --force-ddp --no-sml --no-legs --no-cpu-baseline --full-json $out/bench_rccl_1rank_full.json > $out/bench_rccl_1rank.json 2>> $out/bench.err; cut -c1-200 $out/bench_rccl_1rank.json
timeout 600 python bench.py --gpus 1 --steps 50 --config3 --no-sml --no-legs --no-cpu-baseline --full-json $out/bench_config3_1rank_full.json > $out/bench_config3_1rank.json 2>> $out/bench.err; cut -c1-200 $out/bench_config3_1rank.json
