#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06_f; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -x -s -k "autograph or aliasing or graphed" > $out/gpu_tests.log 2>&1; tail -30 $out/gpu_tests.log | cut -c1-300
for ag in "" "--autograph"; do for wl in caller_rcnet caller_sml; do
  timeout 300 python bench.py --workload $wl $ag --steps 30 --no-children --no-cpu-baseline --full-json $out/$wl$ag.json 2>$out/$wl$ag.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$wl $ag', d['value'], d['ms_per_step'], d['launch_mode'][:60])" || tail -5 $out/$wl$ag.err
done; done
