#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) over tools/bench_head.py: issue / wait counters, then fetch and write sizes; $1 = outdir under gpurun_out
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_head}; mkdir -p $out
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/p1 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_head.py 240 240 100 quick > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p4 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_head.py 240 240 100 quick > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/p2 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_head.py 240 240 100 quick > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/p3 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_head.py 240 240 100 quick > /dev/null 2>&1
rm -f $out/p*/*kernel_trace.csv $out/p*/*agent_info.csv
cd $GRAFT_REPO_ROOT; python3 tools/pmc_summary.py $out
