#!/bin/bash
# PMC passes over a short eager RC-Net bench run, summarised for kernels matching $1 (regex); $2 = out dir under gpurun_out; $3 = extra bench args
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$2; mkdir -p $out
cd /tmp
B="python3 $root/bench.py --steps 2 --warmup 1 --eager --no-sml --no-cpu-baseline --settle-seconds 0 $3"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/p1 -o p -- $B > $out/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p2 -o p -- $B > $out/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $out/p3 -o p -- $B > $out/p3.log 2>&1
cd $root
python3 tools/pmc_summary.py $out | grep -E -A24 "$1" | head -120
