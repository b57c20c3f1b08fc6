"""Per-kernel means of the counters collected by tools/pmc_conv.sh (rd:: kernels only)."""
import csv, glob, os, re, sys
out = sys.argv[1]
agg = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        k = row["Kernel_Name"]
        if "rd::" not in k:
            continue
        name = re.sub(r"\(.*", "", k).replace("void rd::", "")[:60]
        d = agg.setdefault(name, {})
        c = d.setdefault(row["Counter_Name"], [0.0, 0])
        c[0] += float(row["Counter_Value"]); c[1] += 1
        d["_grid"] = row["Grid_Size"]; d["_vgpr"] = row["VGPR_Count"]; d["_lds"] = row["LDS_Block_Size"]
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for name, d in sorted(agg.items()):
        line = "%s grid=%s vgpr=%s lds=%s\n" % (name, d.get("_grid"), d.get("_vgpr"), d.get("_lds"))
        for c, (s, n) in sorted((k, v) for k, v in d.items() if not k.startswith("_")):
            line += "    %-28s %14.0f  (mean of %d dispatches)\n" % (c, s / n, n)
        fo.write(line); print(line, end="")
