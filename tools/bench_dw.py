"""Micro-benchmark of the depthwise kernels on the EfficientNet-Lite3 backbone's shapes (SML, B = 16, 256 x 512): forward, data gradient and
weight gradient through the C ABI, against the HBM time of the tensors each one must move."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riders_amd import engine
from riders_amd.engine import _p, L, _stream
dev = torch.device("cuda:0")
lib = L()
B = int(os.environ.get("BD_B", "16"))
dt, tdt = (1, torch.bfloat16) if os.environ.get("BD_DT", "bf16") == "bf16" else (0, torch.float32)
es = 2 if dt else 4
# (C, k, s, H, W) input maps of every depthwise layer, one line per distinct shape, with its multiplicity in the net
ARCH = [("ds", 1, 3, 1, 24), ("ir", 3, 3, 2, 32), ("ir", 3, 5, 2, 48), ("ir", 5, 3, 2, 96), ("ir", 5, 5, 1, 136), ("ir", 6, 5, 2, 232), ("ir", 1, 3, 1, 384)]


def shapes(H0=256, W0=512):
    out, cin, H, W = {}, 32, H0 // 2, W0 // 2
    for typ, rep, k, s, cout in ARCH:
        for i in range(rep):
            st = s if i == 0 else 1
            C = cin if typ == "ds" else cin * 6
            key = (C, k, st, H, W)
            out[key] = out.get(key, 0) + 1
            H, W = math.ceil(H / st), math.ceil(W / st)
            cin = cout
    return out


ONLY = os.environ.get("BD_ONLY")      # e.g. "288,5,1": that (C, k, s) shape only (PMC passes)
ITERS = int(os.environ.get("BD_ITERS", "50"))
if not ONLY:      # bring the clocks up before the first measurement
    _a = torch.randn(4096, 4096, device=dev)
    for _ in range(40):
        _a = (_a @ _a).clamp_(-1, 1)
    torch.cuda.synchronize()


def timeit(fn, iters=ITERS):
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "ideal": 0.0}
for (C, k, s, H, W), mult in shapes().items():
    if ONLY and ONLY != "%d,%d,%d" % (C, k, s):
        continue
    OH, OW = math.ceil(H / s), math.ceil(W / s)
    pt = max((OH - 1) * s + k - H, 0) // 2
    x = torch.randn((B, H, W, C), device=dev).to(tdt)
    w = torch.randn((C, 1, k, k), device=dev)
    y = torch.empty((B, OH, OW, C), dtype=tdt, device=dev)
    dy = torch.randn((B, OH, OW, C), device=dev).to(tdt)
    dx = torch.empty_like(x)
    dw = torch.zeros_like(w)
    rows = lib.rd_dw_rows(B * OH * OW, C)
    part = torch.empty((rows, C, k * k), dtype=torch.float32, device=dev)
    st = _stream(x)
    srows = lib.rd_dwconv_stats_rows(B, OH, OW, C, k, s) if os.environ.get("BD_STATS") else 0      # BD_STATS=1: the forward the model runs (fused BatchNorm statistics)
    if srows > 0:
        stats = torch.empty((srows, C, 2), dtype=torch.float32, device=dev)
        f = timeit(lambda: lib.rd_dwconv_fwd_stats(_p(x), _p(w), _p(y), _p(stats), B, H, W, C, OH, OW, k, s, pt, dt, st))
    else:
        f = timeit(lambda: lib.rd_dwconv_fwd(_p(x), _p(w), _p(y), B, H, W, C, OH, OW, k, s, pt, dt, st))
    g = timeit(lambda: lib.rd_dwconv_dgrad(_p(dy), _p(w), _p(dx), B, H, W, C, OH, OW, k, s, pt, dt, st))
    wg = timeit(lambda: lib.rd_dwconv_wgrad(_p(x), _p(dy), _p(part), _p(dw), 0, B, H, W, C, OH, OW, k, s, pt, dt, st))
    ideal = (x.numel() + y.numel()) * es / 6.0e6      # us at ~6 TB/s attainable
    print("C=%4d k%d s%d %3dx%3d x%d  fwd %6.1f  dgrad %6.1f  wgrad %6.1f us   (HBM ~%5.1f us)" % (C, k, s, H, W, mult, f, g, wg, ideal), flush=True)
    for n, v in (("fwd", f), ("dgrad", g), ("wgrad", wg), ("ideal", ideal)):
        tot[n] += v * mult
print("per step: fwd %.0f  dgrad %.0f  wgrad %.0f us; HBM-time of one pass %.0f us" % (tot["fwd"], tot["dgrad"], tot["wgrad"], tot["ideal"]))
