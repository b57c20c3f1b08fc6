#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06_c; mkdir -p $out
cd $GRAFT_REPO_ROOT
K="ntu or convergence or aliasing or rccl or graphed or native or adam or pack_batch" bash tools/r06_check.sh c sometests noprof
for v in 0 1 0 1; do
  python bench.py --workload rcnet --steps 100 --no-children --no-cpu-baseline --opts pack_vec=$v --full-json $out/ab_rc_$v.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('rcnet pack_vec=$v', d['value'], d['ms_per_step'])"
  python bench.py --workload sml --steps 100 --no-children --no-cpu-baseline --opts pack_vec=$v --full-json $out/ab_sml_$v.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('sml pack_vec=$v', d['value'], d['ms_per_step'])"
done
