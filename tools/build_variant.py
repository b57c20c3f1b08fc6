"""Build tools/ab/lib_<name>.so: the product library with extra -D flags on selected translation units (A/B and probe builds; the
.so files are git-ignored and travel to the GPU box with the snapshot).

  python3 tools/build_variant.py base rd_wgrad3x3.hip -DRD_WGRAD_SHARE=0
  RIDERS_HIP_LIB=$PWD/tools/ab/lib_base.so python3 tools/bench_conv.py ...      (or tools/ab_wgrad.sh base)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riders_amd import build as b

name, unit, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build(verbose=False)
os.makedirs(os.path.join(ROOT, "tools", "ab"), exist_ok=True)
objs = []
for src, obj, extra in b.units():
    if os.path.basename(src) in unit.split(","):      # several translation units: comma separated
        o2 = os.path.join("/tmp", os.path.basename(obj) + "." + name + ".o")
        subprocess.run([b.HIPCC, "-x", "hip"] + b.FLAGS + extra + defs + ["-c", src, "-o", o2], check=True)
        objs.append(o2)
    else:
        objs.append(obj)
out = os.path.join(ROOT, "tools", "ab", "lib_%s.so" % name)
subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
