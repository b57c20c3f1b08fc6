#!/bin/bash
# round 6, call A: where the bf16 gradient error comes from (tools/grad_localise.py), the one-rank RCCL test (rd_comm.cpp loader change), a short bench line
out=$GRAFT_REPO_ROOT/gpurun_out/r06_a; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python tools/grad_localise.py --out $out/grad_localise.txt > $out/grad_localise.log 2>&1; tail -45 $out/grad_localise.log
timeout 600 python -m pytest tests -q -m gpu -x -k "rccl or native_library" > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
timeout 600 python bench.py --gpus 1 --steps 100 --no-legs --no-cpu-baseline --no-live-traffic --full-json $out/bench_full.json --detail $out/per_shape.txt > $out/bench.json 2> $out/bench.err
tail -2 $out/bench.err | cut -c1-300; cut -c1-600 $out/bench.json
