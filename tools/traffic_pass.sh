#!/bin/bash
# HBM traffic of every kernel family of the bench step from PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE
# in SEPARATE rocprofv3 passes (they do not fit one pass), --kernel-trace only, same command as the bench (eager launches, 3 steps).
# Usage (on the GPU box, from the repo root):  bash tools/traffic_pass.sh rcnet|sml [extra bench args]
# Writes gpurun_out/traffic_<workload>/{fetch,write}/*counter_collection.csv and merges the per-family bytes into profiles/r05_traffic.json.
set -u
wl=${1:-rcnet}; shift || true
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/traffic_$wl; mkdir -p $out
cd /tmp
args="--workload $wl --eager --steps 2 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-sml $*"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o p -- python3 $root/bench.py $args > $out/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o p -- python3 $root/bench.py $args > $out/write.log 2>&1
cd $root
python3 tools/traffic_aggregate.py $wl $out 3 "$*"
