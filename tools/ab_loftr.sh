#!/bin/bash
# A/B of the fused LoFTR layer kernels: rocprofv3 kernel durations of tools/loftr_prof.py (LOFTR_PLAIN=1: no stamps) with
# tools/ab/lib_<name>.so ... and with the current build ("new").  usage: tools/ab_loftr.sh name1 name2 ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp LOFTR_PLAIN=1 LOFTR_REPS=50
out=$GRAFT_REPO_ROOT/gpurun_out/ab_loftr; mkdir -p $out
run() {  # name lib N
  if [ -n "$2" ]; then export RIDERS_HIP_LIB=$2; else unset RIDERS_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$1_$3 -o p -- python3 $GRAFT_REPO_ROOT/tools/loftr_prof.py $3 21 > $out/$1_$3.log 2>&1
  f=$(find $out/$1_$3 -name "p_kernel_stats.csv" | head -1)
  echo "$1 N=$3: $(grep loftr_layer $f | awk -F, '{printf "%s calls=%s avg=%.2f us; ", substr($1,11,18), $2, $4/1000}')"
}
for rep in 1; do
  for n in "$@"; do for N in 240 480; do run $n $GRAFT_REPO_ROOT/tools/ab/lib_$n.so $N; done; done
  for N in 240 480; do run new "" $N; done
done
