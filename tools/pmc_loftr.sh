#!/bin/bash
# PMC passes over the fused LoFTR layer micro-benchmark (product library, 2 repetitions); $1 = outdir under the repo root
export TMPDIR=/tmp LOFTR_PLAIN=1 LOFTR_REPS=2
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
cd /tmp
timeout 100 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $out/p2 -o p -- python3 $GRAFT_REPO_ROOT/tools/loftr_prof.py > $out/p2.log 2>&1
timeout 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/p3 -o p -- python3 $GRAFT_REPO_ROOT/tools/loftr_prof.py > $out/p3.log 2>&1
timeout 100 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $out/p5 -o p -- python3 $GRAFT_REPO_ROOT/tools/loftr_prof.py > $out/p5.log 2>&1
rm -f $out/p*/*kernel_trace.csv $out/p*/*agent_info.csv
