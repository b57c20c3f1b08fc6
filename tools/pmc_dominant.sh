#!/bin/bash
# MFMA-busy / wait counters of every named kernel of the bench step on the DEFAULT build (north_star: "rocprof-reported ... MFMA utilisation"):
# one rocprofv3 --pmc pass (SQ counters + GRBM_GUI_ACTIVE, --kernel-trace only) over 2 eager steps of the bench command, aggregated per kernel
# instantiation into profiles/r05_pmc_dominant.json, which bench.py attaches to its `roofline` object for the dominant kernel.
# Usage (GPU box, repo root):  bash tools/pmc_dominant.sh rcnet|sml
set -u
wl=${1:-rcnet}; shift || true
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_dom_$wl; mkdir -p $out
cd /tmp
args="--workload $wl --eager --steps 2 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-sml $*"
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p -o p -- python3 $root/bench.py $args > $out/p.log 2>&1
cd $root
python3 tools/pmc_dominant.py $wl $out
rm -rf $out/p
