"""Aggregate the counter pass of tools/pmc_dominant.sh per kernel instantiation -> profiles/r05_pmc_dominant.json.
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the fraction of the chip's matrix-pipe cycles that were busy
while the kernel ran.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs (16 per 16x16x32 bf16 MFMA: 47.2 M for 2.95 M
instructions); rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (958 K for a ~50 us dispatch at ~2.3 GHz = 8 x 120 K), hence the / 8.  wait fractions are of SQ_WAVE_CYCLES (quad-cycles; WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall)."""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from traffic_aggregate import csrc_fingerprint, kernel_key  # noqa: E402


def main():
    wl, out = sys.argv[1], sys.argv[2]
    agg = {}
    for f in glob.glob(os.path.join(out, "p", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if "rd::" not in row["Kernel_Name"] and "rd_f16::" not in row["Kernel_Name"]:
                continue
            d = agg.setdefault(kernel_key(row["Kernel_Name"]), {})
            c = d.setdefault(row["Counter_Name"], [0.0, 0])
            c[0] += float(row["Counter_Value"]); c[1] += 1
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "r05_pmc_dominant.json")
    db = json.load(open(path)) if os.path.exists(path) else {}
    db[wl] = {}
    for k, d in agg.items():
        m = {c: v[0] / v[1] for c, v in d.items()}
        n = max(v[1] for v in d.values())
        e = dict(workload=wl, dispatches=n, per_dispatch={c: m[c] for c in sorted(m)}, csrc_sha1=csrc_fingerprint(root),
                 source="rocprofv3 --pmc (tools/pmc_dominant.sh), mean over the dispatches of 3 eager steps")
        if m.get("GRBM_GUI_ACTIVE"):
            e["mfma_util"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
        if m.get("SQ_WAVE_CYCLES"):
            e["wait_any_frac"] = m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
            e["wait_inst_frac"] = m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
            e["active_inst_frac"] = m.get("SQ_ACTIVE_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
        db[wl][k] = e
    json.dump(db, open(path, "w"), indent=1, sort_keys=True)
    top = sorted(((v.get("mfma_util", 0.0), k) for k, v in db[wl].items() if v.get("per_dispatch", {}).get("SQ_INSTS_MFMA", 0) > 0), reverse=True)[:8]
    print("pmc:", [(k[:50], round(u, 3)) for u, k in top])


if __name__ == "__main__":
    main()
