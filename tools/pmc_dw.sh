#!/bin/bash
# PMC passes for one shape of tools/bench_dw.py: $1 = "C,k,s", $2 = out dir under gpurun_out
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$2; mkdir -p $out
cd /tmp
export BD_ONLY=$1 BD_ITERS=5
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/p1 -o p -- python3 $root/tools/bench_dw.py > $out/p1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p2 -o p -- python3 $root/tools/bench_dw.py > $out/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $out/p3 -o p -- python3 $root/tools/bench_dw.py > $out/p3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $out/p4 -o p -- python3 $root/tools/bench_dw.py > $out/p4.log 2>&1
cd $root
python3 tools/pmc_summary.py $out
