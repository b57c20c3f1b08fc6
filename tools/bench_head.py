"""Micro-benchmark of the fused decoder-head kernels (rd_head.hip) at RC-Net's RoI geometry: every kernel by itself, per option set
(tile size of the forward kernel, channels per work item), next to the HBM time of its algorithmic bytes.  GPU box only.
usage: python tools/bench_head.py [R=240] [H=240] [W=100]"""
import ctypes
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riders_amd import _lib, engine   # noqa: E402


def p(t):
    return ctypes.c_void_p(t.data_ptr())


def timeit(fn, it=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def main():
    quick = "quick" in sys.argv      # one option set, bf16 only (the PMC passes)
    argv = [v for v in sys.argv[1:] if v != "quick"]
    R, H, W = [int(v) for v in (argv[:3] + ["240", "240", "100"][len(argv):])]
    lib = _lib.load()
    dev = torch.device("cuda:0")
    C = 16
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for dt_name, tdt, dt in (("bf16", torch.bfloat16, 1), ("fp32", torch.float32, 0))[:1 if quick else 2]:
        torch.manual_seed(0)
        y = torch.randn(R, H, W, C, device=dev).to(tdt)
        dl = (torch.randn(R, H, W, 1, device=dev) * 1e-3).to(tdt)
        coef = torch.randn(4, C, device=dev)
        coef[3].abs_()
        w = torch.randn(1, C, 3, 3, device=dev) * 0.1
        logits = torch.empty(R, H, W, 1, device=dev, dtype=tdt)
        dy = torch.empty_like(y)
        rows = lib.rd_bn_head_rows(R, H, W)
        part = torch.empty(rows, 88, 2, device=dev)
        coef2 = torch.empty(2, C, device=dev)
        dg, db, dw = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(1, C, 3, 3, device=dev)
        nb = y.numel() * y.element_size()
        px = R * H * W * y.element_size()
        print("%s  R=%d %dx%d  tensor %.1f MB  (HBM time at 5 TB/s: fwd %.0f us, sums %.0f us, apply %.0f us)" %
              (dt_name, R, H, W, nb / 1e6, (nb + px) / 5e6, (nb + px) / 5e6, (2 * nb + px) / 5e6))
        sets = ((1360, 0x444), (1024, 0x444), (1800, 0x444), (700, 0x444), (1360, 0x888), (1024, 0x888))
        for np_, cpi in (sets[1:2] if quick else sets):
            engine.set_option("head_np", np_)
            engine.set_option("head_cpi", cpi)
            f = timeit(lambda: lib.rd_bn_head_fwd(p(y), p(coef[0]), p(coef[1]), 2, 0.2, p(w), p(logits), R, H, W, C, dt, st))
            r = timeit(lambda: lib.rd_bn_head_bwd_reduce(p(dl), p(y), p(coef[2]), p(coef[3]), p(coef[0]), p(coef[1]), 2, 0.2, p(w), p(part), R, H, W, C, dt, st))
            a = timeit(lambda: lib.rd_bn_head_bwd_apply(p(dl), p(y), p(coef[2]), p(coef[3]), p(coef[0]), p(coef[1]), 2, 0.2, p(w), p(part), rows, p(coef2), p(dg), p(db),
                                                         0, p(dw), 0, p(dy), R, H, W, C, dt, st))
            print("  head_np %4d  cpi %03x   fwd %6.1f us   sums + head wgrad %6.1f us   finalize + apply %6.1f us" % (np_, cpi, f, r, a))
        engine.set_option("head_np", None)
        engine.set_option("head_cpi", None)


if __name__ == "__main__":
    main()
