#!/bin/bash
# kernel-trace profile of the SML training step: $1 = out dir under gpurun_out
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$1; mkdir -p $out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o sml -- python3 $root/bench.py --steps 15 --warmup 3 --no-cpu-baseline --workload sml --settle-seconds 0 > $out/bench.log 2>&1
cd $root
tail -n 1 $out/bench.log | cut -c1-400
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = float(max(1, sum(int(r["Calls"]) for r in rows if "sml_loss_finalize" in r["Name"])))
print("kernel time per step: %.2f ms, launches per step: %.0f" % (tot / steps / 1e6, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:32]:
    print("%-86s %5d %8.1f us  %5.2f%%  %6.3f ms/step" % (r["Name"][:86], int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / steps / 1e6))
PY
