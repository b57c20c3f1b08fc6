#!/bin/bash
# PMC passes over the fused LoFTR layer micro-benchmark (product library); $1 = outdir under the repo root
export TMPDIR=/tmp LOFTR_PLAIN=1
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/p1 -o p -- python3 $GRAFT_REPO_ROOT/tools/loftr_prof.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $out/p2 -o p -- python3 $GRAFT_REPO_ROOT/tools/loftr_prof.py > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/p3 -o p -- python3 $GRAFT_REPO_ROOT/tools/loftr_prof.py > $out/p3.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --kernel-trace --output-format csv -d $out/p4 -o p -- python3 $GRAFT_REPO_ROOT/tools/loftr_prof.py > $out/p4.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $out/p5 -o p -- python3 $GRAFT_REPO_ROOT/tools/loftr_prof.py > $out/p5.log 2>&1
rm -f $out/p*/*kernel_trace.csv $out/p*/*agent_info.csv
