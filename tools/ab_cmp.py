"""Tabulate an A/B log written by tools/ab_*.sh: python3 tools/ab_cmp.py gpurun_out/ab/x.txt"""
import re, sys
d = {}
for l in open(sys.argv[1]):
    m = re.match(r'(base|new) (fwd|dgrad|wgrad) (NHW=\S+ Cin=\d+ Cout=\d+) \w+: ([\d.]+) ms', l)
    if m:
        d.setdefault((m.group(3), m.group(2)), {})[m.group(1)] = float(m.group(4))
for k, v in d.items():
    print("%-36s %-6s base %.3f  new %.3f  speedup %.3f" % (k[0], k[1], v.get('base', 0), v.get('new', 0), v.get('base', 0) / max(v.get('new', 1), 1e-9)))
