#!/bin/bash
# Whole-step bench lines and per-launch-shape lines of the product library against tools/ab/lib_<name>.so variants.
# usage: tools/ab_lib.sh "grep pattern" workload name1 [name2 ...]     ("new" = the product library)
cd $GRAFT_REPO_ROOT
pat=$1; wl=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/ab_lib; mkdir -p $out
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = "new" ]; then unset RIDERS_HIP_LIB; else export RIDERS_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_$n.so; fi
  timeout 600 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-legs --no-sml --workload $wl --detail $out/${n}_$wl.txt 2>$out/err.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$n $wl', '%.1f img/s %.3f ms'%(d['value'], d['ms_per_step']))"
  [ $rep = 2 ] && grep -h -E "$pat" $out/${n}_$wl.txt* | head -8 | cut -c1-150
done
done
