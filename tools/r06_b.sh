#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06_b; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python tools/grad_localise.py --out $out/grad_localise.txt > $out/grad_localise.log 2>&1; tail -36 $out/grad_localise.log | cut -c1-220
K="ntu or convergence or aliasing or rccl or graphed or native or adam or pack" bash tools/r06_check.sh b sometests noprof
