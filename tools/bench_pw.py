"""Micro-benchmark of the Scale Map Learner's 1x1 (pointwise) convolutions on its deep stages (B = 16, 288x576 padded input: 9x18 ... 72x144
maps) through the C ABI: forward with / without the BatchNorm statistics epilogue, data gradient with / without the addend, per shape, with
the kernel instantiation the library routes it to.  Inputs rotate over eight buffer sets so that a launch does not find its operands in L2.
`python tools/bench_pw.py [mode ...]`, a mode = comma-separated routing options as in tools/bench_conv.py."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riders_amd import engine
from riders_amd.engine import _desc, _p, L, _stream
dev = torch.device("cuda:0")
lib = L()
lib.rd_conv_fwd_kernel_name.restype = ctypes.c_char_p
dt, tdt = 1, torch.bfloat16
# (N, H, W, Cin, Cout, launches per step as forward, as data gradient of the TRANSPOSED layer)
import os
SHAPES = [(16, 18, 36, 136, 816), (16, 18, 36, 816, 136), (16, 9, 18, 1392, 232), (16, 9, 18, 232, 1392), (16, 18, 36, 96, 576), (16, 18, 36, 576, 96),
          (16, 36, 72, 48, 288), (16, 36, 72, 288, 48), (16, 9, 18, 1392, 384), (16, 72, 144, 32, 192), (16, 72, 144, 192, 32), (16, 18, 36, 576, 136),
          (16, 9, 18, 816, 232), (16, 9, 18, 512, 256), (16, 18, 36, 256, 128)]
if os.environ.get("PW_ONLY"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["PW_ONLY"].split(",")]
NB = 8


def timeit(fn, iters=40):
    for i in range(NB):
        assert fn(i) == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i % NB)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def run(tag):
    tot = [0.0, 0.0, 0.0]
    for (N, H, W, Cin, Cout) in SHAPES:
        M = N * H * W
        xs = [torch.randn((N, H, W, Cin), device=dev).to(tdt) for _ in range(NB)]
        ys = [torch.empty((N, H, W, Cout), dtype=tdt, device=dev) for _ in range(NB)]
        adds = [torch.randn((N, H, W, Cout), device=dev).to(tdt) for _ in range(NB)]
        w = torch.nn.Parameter(torch.randn((Cout, Cin, 1, 1), device=dev) * 0.05)
        d = _desc(dt, N, H, W, Cin, 0, False, H, W, Cout, 1, 1, 1, 0, 1, H, W, 0, 0.0, Cout)
        wp = engine.packed_weight(w, 0, dt)
        rows = lib.rd_conv_stats_rows(ctypes.byref(d))
        stats = torch.empty((rows, Cout, 2), dtype=torch.float32, device=dev)
        st = _stream(xs[0])
        name = lib.rd_conv_fwd_kernel_name(ctypes.byref(d)).decode()
        t0 = timeit(lambda i: lib.rd_conv_fwd(ctypes.byref(d), _p(xs[i]), None, _p(wp), None, _p(ys[i]), None, None, st))
        t1 = timeit(lambda i: lib.rd_conv_fwd(ctypes.byref(d), _p(xs[i]), None, _p(wp), None, _p(ys[i]), None, _p(stats), st))
        t2 = timeit(lambda i: lib.rd_conv_fwd_add(ctypes.byref(d), _p(xs[i]), None, _p(wp), None, _p(adds[i]), _p(ys[i]), st)) if lib.rd_conv_add_ok(ctypes.byref(d)) else float("nan")
        fl = 2.0 * M * Cin * Cout
        by = M * (Cin + Cout) * 2
        print("%s M=%6d %4d->%4d  plain %6.1f us (%5.0f TFLOP/s %5.0f GB/s)  +stats %6.1f us (rows %4d)  +addend %6.1f us   %s" %
              (tag, M, Cin, Cout, t0, fl / t0 / 1e6, by / t0 / 1e3, t1, rows, t2, name.replace("_kernel", "").replace("rd::bf16_t", "bf16")), flush=True)
        tot[0] += t0; tot[1] += t1; tot[2] += t2
    print("%s totals plain %.1f  +stats %.1f  +addend %.1f us" % (tag, tot[0], tot[1], tot[2]), flush=True)


for mode in (sys.argv[1:] or ["default"]):
    lib.rd_clear_options()      # a mode starts from the defaults
    for kv in mode.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            try:
                engine.set_option(k, None if v == "" else int(v))
            except (RuntimeError, ValueError):
                pass
    run("[%s]" % mode)
