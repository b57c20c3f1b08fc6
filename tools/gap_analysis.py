"""GPU idle time between kernels in the steady state of a bench run, from a rocprofv3 --kernel-trace CSV:
   python3 tools/gap_analysis.py <dir with *_kernel_trace.csv> [adam-kernel-substring]
Steps are delimited by the optimizer launches; 10 complete steps ending `skip` (default 4) steps before the end are analysed (busy = union of kernel intervals)."""
import csv, glob, sys
d = sys.argv[1]; key = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if key in r[2]]
if len(marks) < 16:
    print("too few steps", len(marks)); sys.exit(0)
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 4      # the bench ends with 3 instrumented eager steps
lo, hi = marks[-11 - skip], marks[-1 - skip]
seg = rows[lo + 1:hi + 1]
t0, t1 = seg[0][0], seg[-1][1]
busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
gaps = []
for s, e, _ in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
n = 10
print("steps %d  wall %.3f ms/step  busy %.3f ms/step (%.1f %%)  launches/step %.0f  idle gaps: n=%d mean %.2f us, >5us: %d (%.3f ms/step)" % (
    n, (t1 - t0) / 1e6 / n, busy / 1e6 / n, 100.0 * busy / (t1 - t0), len(seg) / n, len(gaps), sum(gaps) / max(1, len(gaps)) / 1e3,
    sum(1 for g in gaps if g > 5000), sum(g for g in gaps if g > 5000) / 1e6 / n))
