"""Phase timing of the fused LoFTR layer kernels (RD_LOFTR_PROF build of rd_loftr.hip, wall_clock64 stamps by thread 0 of each workgroup).

  python3 tools/loftr_prof.py build      # here (hipcc cross-compiles): tools/ab/libriders_hip_prof.so
  python3 tools/loftr_prof.py [N L]      # on the GPU box
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "tools", "ab", "libriders_hip_prof.so")

sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from riders_amd import build
    os.makedirs(os.path.dirname(PROF), exist_ok=True)
    obj = os.path.join(ROOT, "tools", "ab", "rd_loftr_prof.o")
    subprocess.check_call([build.HIPCC, "-x", "hip"] + build.FLAGS + ["-DRD_LOFTR_PROF", "-I", os.path.join(ROOT, "include"), "-c",
                           os.path.join(build.CSRC, "rd_loftr.hip"), "-o", obj])
    objs = [obj if o.endswith("rd_loftr.hip.o") else o for (_, o, _) in build.units()]      # both precision builds; the bf16 / fp32 LoFTR unit replaced
    subprocess.check_call([build.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROF] + objs)
    print(PROF)
    sys.exit(0)

PLAIN = os.environ.get("LOFTR_PLAIN") == "1"   # product library, no stamps (for rocprofv3 --pmc runs)
if not PLAIN:
    os.environ["RIDERS_HIP_LIB"] = PROF
sys.path.insert(0, ROOT)
import torch
from riders_amd import engine, _lib
from riders_amd.linear_attention import LoFTREncoderLayer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 240
L = int(sys.argv[2]) if len(sys.argv) > 2 else 21
dev = torch.device("cuda:0")
lib = _lib.load()
DT = torch.float32 if os.environ.get("LOFTR_DTYPE") == "fp32" else torch.bfloat16
engine.set_compute_dtype("fp32" if DT == torch.float32 else "bf16")
raw = None if PLAIN else ctypes.CDLL(PROF)
layer = LoFTREncoderLayer(128, 8).to(dev)   # fp32 master parameters, bf16 activations
names = {0: "load x/src", 1: "q,k,v GEMMs", 2: "attention", 3: "load att", 4: "merge GEMM", 5: "norm1", 6: "mlp0 GEMM", 7: "mlp2 GEMM", 8: "norm2+out",
         10: "norm2 bwd", 11: "dhid GEMM", 12: "dcat GEMM", 13: "norm1 bwd", 14: "datt GEMM", 15: "attention bwd", 16: "dq Wq", 17: "dk Wk", 18: "dv Wv",
         19: "store"}
for cross in ((False,) if PLAIN else (False, True)):
    x = torch.randn(N, L, 128, device=dev, dtype=DT).requires_grad_()
    s = torch.randn(N, L, 128, device=dev, dtype=DT).requires_grad_()
    reps = int(os.environ.get("LOFTR_REPS", "20"))
    warm = 3 if reps > 2 else 0
    for it in range(reps + warm):
        if it == warm:
            torch.cuda.synchronize()
            if raw: raw.rd_debug_loftr_prof(None, 1)
        o = layer(x, s if cross else x)
        o.sum().backward()
    torch.cuda.synchronize()
    if PLAIN:
        continue
    buf = (ctypes.c_ulonglong * 32)()
    assert raw.rd_debug_loftr_prof(buf, 1) == 0
    print("cross" if cross else "self", "N=%d L=%d: mean per workgroup, us (100 MHz wall clock)" % (N, L))
    for i in sorted(names):
        print("  %-14s %7.2f" % (names[i], buf[i] / (reps * N) / 100.0))
    print("  fwd total %.2f   bwd total %.2f" % (sum(buf[i] for i in range(10)) / (reps * N) / 100.0, sum(buf[i] for i in range(10, 20)) / (reps * N) / 100.0))
