#!/bin/bash
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out; cd /tmp
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/p1 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_roi.py 0 > $out/p1.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/p4 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_roi.py 0 > $out/p4.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $out/p2 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_roi.py 0 > $out/p2.log 2>&1
rm -f $out/p*/*kernel_trace.csv $out/p*/*agent_info.csv
