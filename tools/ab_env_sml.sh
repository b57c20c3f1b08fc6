#!/bin/bash
# SML step with one environment variable at several values: whole-step img/s and the per-launch-shape lines matching a pattern.
# usage: tools/ab_env_sml.sh VAR "v1 v2 v3" "grep pattern" [workload]
cd $GRAFT_REPO_ROOT
var=$1; vals=$2; pat=$3; wl=${4:-sml}
out=$GRAFT_REPO_ROOT/gpurun_out/ab_env; mkdir -p $out
for v in $vals; do
  env $var=$v timeout 600 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-legs --no-sml --workload $wl --detail $out/${var}_$v.txt 2>$out/err.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$var=$v', '%.1f img/s %.3f ms'%(d['value'], d['ms_per_step']))"
  grep -h -E "$pat" $out/${var}_$v.txt* | head -12 | cut -c1-150
done
