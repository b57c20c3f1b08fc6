#!/bin/bash
# PMC passes for one weight-gradient shape of tools/bench_conv.py on two builds: $1 = shape index, $2 = out dir under gpurun_out, $3.. = lib names under tools/ab ("cur" = current build)
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
idx=$1; outn=$2; shift; shift
export BC_ONLY=wgrad:$idx BC_SKIP_CONV=1
for n in "$@"; do
  out=$root/gpurun_out/$outn/$n; mkdir -p $out
  if [ "$n" = "cur" ]; then unset RIDERS_HIP_LIB; else export RIDERS_HIP_LIB=$root/tools/ab/lib_$n.so; fi
  cd /tmp
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/p1 -o p -- python3 $root/tools/bench_conv.py RD_X=$n > $out/p1.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/p2 -o p -- python3 $root/tools/bench_conv.py RD_X=$n > $out/p2.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $out/p3 -o p -- python3 $root/tools/bench_conv.py RD_X=$n > $out/p3.log 2>&1
  cd $root
  python3 tools/pmc_summary.py $out > /dev/null 2>&1
  rm -rf $out/p1 $out/p2 $out/p3
done
