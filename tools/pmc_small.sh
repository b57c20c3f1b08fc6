#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) for one layer micro-benchmark: $1=NHW $2=Cin $3=Cout $4=dtype $5=fwd|wgrad $6=outdir
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/$6; mkdir -p $out
cd /tmp
RD_NHW=$1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/p1 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_wgrad.py $2 $3 $4 $5 > /dev/null 2>&1
RD_NHW=$1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/p2 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_wgrad.py $2 $3 $4 $5 > /dev/null 2>&1
RD_NHW=$1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/p3 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_wgrad.py $2 $3 $4 $5 > /dev/null 2>&1
rm -f $out/p*/*kernel_trace.csv $out/p*/*agent_info.csv
