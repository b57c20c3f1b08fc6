#!/bin/bash
# rocprofv3 kernel durations of the depthwise micro-benchmark for some (C,k,s) shapes: tools/prof_dw.sh "1392,5,1" "816,5,1" ...
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for sh in "$@"; do
  rm -rf /tmp/pdw
  BD_ONLY=$sh rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pdw -o p -- python3 $root/tools/bench_dw.py > /tmp/pdw.log 2>&1
  echo "== $sh"
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/pdw/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print("%-70s calls %5s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
