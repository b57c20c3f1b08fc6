"""Merge the FETCH_SIZE / WRITE_SIZE passes of tools/traffic_pass.sh into profiles/r05_traffic.json (bytes per step and kernel family).
Units and corrections as MI355X_MICROARCH.md (HBM section): both counters are in KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of
wide (16 B / lane) streaming reads at 64 B, so it is doubled; WRITE_SIZE is taken as reported (uncalibrated there)."""
import csv
import glob
import json
import os
import re
import sys

FAMILIES = [
    ("dwconv", r"dwconv|dw_run_kernel|dw_dgrad2_kernel|dw_wgrad"),
    ("bn_finalize", r"bn_finalize|bn_bwd_finalize|bn_head_finalize"),
    ("conv_gemm", r"conv3x3_frag_kernel|conv3x3_patch_kernel|conv3x3_small_kernel|conv_gemm_kernel|conv1x1_direct_kernel|conv3x3_c1_kernel|conv_few_kernel|bn_head_fwd_kernel"),
    ("conv_wgrad", r"wgrad"),
    ("loftr_layer", r"loftr_layer"),
    ("bn_apply", r"affine_act"),
    ("bn_backward", r"col_reduce|bn_bwd|bn_head_bwd"),
    ("roi_pool", r"roi_pool"),
    ("pooling", r"maxpool"),
    ("optimizer", r"adam_kernel"),
]


def csrc_fingerprint(root):
    """sha1 over the kernel sources (riders_amd/csrc/*.hip, *.h, *.cpp): identifies the build the counters were taken on."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(root, "riders_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".cpp")):
            h.update(fn.encode()); h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()


def family(name):
    if "rd::" not in name:
        return None
    for fam, pat in FAMILIES:
        if re.search(pat, name):
            return fam
    return "elementwise"


def kernel_key(name):
    """`void rd::conv3x3_frag_kernel<rd::bf16_t, 8, 1, 4, true, true>(rdt::ConvArgs, rd::FragGeom)` -> the name bench.py groups by."""
    k = re.sub(r"\(.*$", "", name).replace("void ", "")
    return re.sub(r"^rd(_f16)?::", "", k)


def read(dirname, counter):
    tot, per_kernel = {}, {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if row["Counter_Name"] != counter:
                continue
            fam = family(row["Kernel_Name"])
            if fam is None:
                continue
            v = float(row["Counter_Value"]) * 1024.0
            tot[fam] = tot.get(fam, 0.0) + v
            short = kernel_key(row["Kernel_Name"])
            e = per_kernel.setdefault(short, [0.0, 0])
            e[0] += v; e[1] += 1
    return tot, per_kernel


def main():
    wl, out, steps, extra = sys.argv[1], sys.argv[2], int(sys.argv[3]), (sys.argv[4] if len(sys.argv) > 4 else "")
    fetch, fk = read(os.path.join(out, "fetch"), "FETCH_SIZE")
    write, wk = read(os.path.join(out, "write"), "WRITE_SIZE")
    line = None
    for ln in open(os.path.join(out, "fetch.log")):
        if ln.startswith("{"):
            line = json.loads(ln)
    cfgkey = "unknown"
    if line is not None:
        m = re.search(r"batch (\d+)/GPU.*?(\d+)x(\d+) (?:image|frames)", line["config"]["workload"])
        cfgkey = "%s_b%s_%sx%s_%s" % (wl, m.group(1), m.group(2), m.group(3), {"f32": "fp32", "bf16": "bf16", "f16": "fp16"}[line["dtype"]])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "r05_traffic.json")
    db = json.load(open(path)) if os.path.exists(path) else {}
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over %d eager steps of `bench.py --workload %s %s` "
            "(tools/traffic_pass.sh); KB -> bytes, FETCH_SIZE doubled (gfx950 counts 128-B streaming requests at 64 B, MI355X_MICROARCH.md), "
            "WRITE_SIZE as reported" % (steps, wl, extra))
    db[cfgkey] = {fam: dict(fetch_bytes_per_step=2.0 * fetch.get(fam, 0.0) / steps, write_bytes_per_step=write.get(fam, 0.0) / steps,
                            bytes_per_step=(2.0 * fetch.get(fam, 0.0) + write.get(fam, 0.0)) / steps, note=note)
                  for fam in sorted(set(fetch) | set(write))}
    db[cfgkey]["_csrc_sha1"] = csrc_fingerprint(root)      # bench.py reports the figures as stale when the kernel sources have changed since
    db[cfgkey]["note"] = note
    kern = {}
    for k in sorted(set(fk) | set(wk)):
        fb, fn = fk.get(k, [0.0, 0]); wb, wn = wk.get(k, [0.0, 0])
        n = max(fn, wn, 1)
        kern[k] = dict(launches_per_step=n / steps, fetch_bytes_per_launch=2.0 * fb / max(fn, 1), write_bytes_per_launch=wb / max(wn, 1),
                       bytes_per_launch=2.0 * fb / max(fn, 1) + wb / max(wn, 1))
    db[cfgkey]["kernels"] = kern
    json.dump(db, open(path, "w"), indent=1, sort_keys=True)
    print("traffic:", cfgkey, {k: round(v["bytes_per_step"] / 1e6, 1) for k, v in db[cfgkey].items() if isinstance(v, dict) and "bytes_per_step" in v}, "MB/step")


if __name__ == "__main__":
    main()
