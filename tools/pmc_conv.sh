#!/bin/bash
# PMC passes for one shape of tools/bench_conv.py: $1 = fwd:<i> | dgrad:<i>, $2 = bench_conv mode list (e.g. "0 1"), $3 = out dir under gpurun_out
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$3; mkdir -p $out
cd /tmp
export BC_ONLY=$1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/p1 -o p -- python3 $root/tools/bench_conv.py $2 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/p2 -o p -- python3 $root/tools/bench_conv.py $2 > $out/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/p3 -o p -- python3 $root/tools/bench_conv.py $2 > $out/p3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/p4 -o p -- python3 $root/tools/bench_conv.py $2 > $out/p4.log 2>&1
cd $root
python3 tools/pmc_summary.py $out
