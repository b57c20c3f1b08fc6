"""Micro-benchmark of the 3x3 convolution kernels on RC-Net's layer shapes (B = 8, R = 240 RoIs), forward and data gradient, through the C
ABI.  A/B inside one process: every argument is a mode = a list of routing options (engine.set_option / rd_set_option), e.g.
`python tools/bench_conv.py default frag_v128=0 frag_v128=,frag32_v128=3`."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riders_amd import engine
from riders_amd.engine import _desc, _p, L, _stream
dev = torch.device("cuda:0")
lib = L()
dt, tdt = 1, torch.bfloat16
SHAPES = [  # N, H, W, Cin, Cout, C1 (first-source channels of a concat input, 0 = single source), up-from (h, w) or None
    (240, 15, 6, 384, 256, 256, None), (240, 15, 6, 256, 256, 0, (7, 3)),
    (240, 30, 12, 256, 128, 128, None), (240, 30, 12, 256, 128, 0, (15, 6)),
    (240, 60, 25, 128, 64, 64, None), (240, 60, 25, 128, 64, 0, (30, 12)),
    (240, 120, 50, 64, 32, 32, None), (240, 120, 50, 64, 32, 0, (60, 25)),
    (8, 124, 153, 64, 64, 0, None), (8, 62, 77, 128, 128, 0, None), (8, 31, 39, 128, 128, 0, None), (8, 16, 20, 128, 128, 0, None),
]
# data gradients that are convolutions with >= 64 input channels: (N, H, W, Cin = fwd Cout, Cout = fwd Cin, D1)
DGRADS = [(240, 15, 6, 256, 384, 256), (240, 15, 6, 256, 256, 256), (240, 30, 12, 128, 256, 128), (240, 60, 25, 64, 128, 64),
          (8, 124, 153, 64, 64, 64), (8, 62, 77, 128, 128, 128), (8, 31, 39, 128, 128, 128), (8, 16, 20, 128, 128, 128)]


def timeit(fn, iters=20):
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


ONLY = os.environ.get("BC_ONLY")      # e.g. "fwd:6" = forward shape index 6 only (PMC passes)
if os.environ.get("BC_ROIS"):          # block-count quantisation study: the decoder shapes with another number of RoIs (default 240)
    _n = int(os.environ["BC_ROIS"])
    SHAPES = [((_n if s[0] == 240 else s[0]),) + tuple(s[1:]) for s in SHAPES]
    DGRADS = [((_n if s[0] == 240 else s[0]),) + tuple(s[1:]) for s in DGRADS]


def run(tag):
    tot = 0.0
    for si, (N, H, W, Cin, Cout, C1, up) in enumerate(SHAPES):
        if (ONLY and ONLY != "fwd:%d" % si) or os.environ.get("BC_SKIP_CONV"):
            continue
        hs, ws = up if up else (H, W)
        c1 = C1 if C1 else Cin
        x1 = torch.randn((N, hs, ws, c1), device=dev).to(tdt)
        x2 = torch.randn((N, hs, ws, Cin - c1), device=dev).to(tdt) if c1 < Cin else None
        w = torch.nn.Parameter(torch.randn((Cout, Cin, 3, 3), device=dev))
        d = _desc(dt, N, H, W, c1, Cin - c1, up is not None, hs, ws, Cout, 3, 3, 1, 1, 1, H, W, 0, 0.0, Cout)
        wp = engine.packed_weight(w, 0, dt)
        y = torch.empty((N, H, W, Cout), dtype=tdt, device=dev)
        rows = lib.rd_conv_stats_rows(ctypes.byref(d))
        stats = torch.empty((rows, Cout, 2), dtype=torch.float32, device=dev)
        st = _stream(x1)
        ms = timeit(lambda: lib.rd_conv_fwd(ctypes.byref(d), _p(x1), _p(x2), _p(wp), None, _p(y), None, _p(stats), st))
        fl = 2.0 * N * H * W * Cin * Cout * 9
        by = (x1.numel() + (x2.numel() if x2 is not None else 0) + y.numel()) * 2
        print("%s fwd   M=%7d %3d->%3d%s  %.3f ms  %6.1f TFLOP/s  %6.1f GB/s" % (tag, N * H * W, Cin, Cout, " up" if up else ("  c" if x2 is not None else "   "), ms, fl / ms / 1e9, by / ms / 1e6), flush=True)
        tot += ms
    for si, (N, H, W, Cin, Cout, D1) in enumerate(DGRADS):
        if (ONLY and ONLY != "dgrad:%d" % si) or os.environ.get("BC_SKIP_CONV"):
            continue
        dy = torch.randn((N, H, W, Cin), device=dev).to(tdt)
        w = torch.nn.Parameter(torch.randn((Cin, Cout, 3, 3), device=dev))      # forward weight (Cout_f = Cin here, Cin_f = Cout here)
        wpd = engine.packed_weight(w, 1, dt)
        dd = _desc(dt, N, H, W, Cin, 0, False, H, W, Cout, 3, 3, 1, 1, 1, H, W, 0, 0.0, D1)
        dx1 = torch.empty((N, H, W, D1), dtype=tdt, device=dev)
        dx2 = torch.empty((N, H, W, Cout - D1), dtype=tdt, device=dev) if D1 < Cout else None
        st = _stream(dy)
        ms = timeit(lambda: lib.rd_conv_fwd(ctypes.byref(dd), _p(dy), None, _p(wpd), None, _p(dx1), _p(dx2), None, st))
        fl = 2.0 * N * H * W * Cin * Cout * 9
        print("%s dgrad M=%7d %3d->%3d     %.3f ms  %6.1f TFLOP/s" % (tag, N * H * W, Cin, Cout, ms, fl / ms / 1e9), flush=True)
        tot += ms
    print("%s total %.3f ms" % (tag, tot), flush=True)
    if os.environ.get("BC_SKIP_WGRAD"):
        return
    wt = 0.0
    for si, (N, H, W, Cin, Cout, C1, up) in enumerate(SHAPES):
        if ONLY and ONLY != "wgrad:%d" % si:
            continue
        hs, ws = up if up else (H, W)
        c1 = C1 if C1 else Cin
        x1 = torch.randn((N, hs, ws, c1), device=dev).to(tdt)
        x2 = torch.randn((N, hs, ws, Cin - c1), device=dev).to(tdt) if c1 < Cin else None
        dy = torch.randn((N, H, W, Cout), device=dev).to(tdt)
        d = _desc(dt, N, H, W, c1, Cin - c1, up is not None, hs, ws, Cout, 3, 3, 1, 1, 1, H, W, 0, 0.0, Cout)
        ws_ = torch.empty(lib.rd_conv_wgrad_workspace_bytes(ctypes.byref(d)) // 4, dtype=torch.float32, device=dev)
        dw = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=dev)
        st = _stream(x1)
        ms = timeit(lambda: lib.rd_conv_wgrad(ctypes.byref(d), _p(x1), _p(x2), _p(dy), _p(ws_), _p(dw), 0, st))
        fl = 2.0 * N * H * W * Cin * Cout * 9
        print("%s wgrad M=%7d %3d->%3d%s  %.3f ms  %6.1f TFLOP/s" % (tag, N * H * W, Cin, Cout, " up" if up else ("  c" if x2 is not None else "   "), ms, fl / ms / 1e9), flush=True)
        wt += ms
    print("%s wgrad total %.3f ms" % (tag, wt), flush=True)


for mode in (sys.argv[1:] or ["default"]):
    # a mode is a comma-separated list of routing options of the library, "frag_v128=0,frag32_v128=3" (rd_set_option; the old RD_FRAG_V128=0
    # spelling is accepted), "name=" clears one; anything that is not an option is only a label
    for kv in mode.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            name = k[3:].lower() if k.startswith("RD_") else k
            try:
                engine.set_option(name, None if v == "" else int(v))
            except (RuntimeError, ValueError):
                pass
    run("[%s]" % mode)
