"""Micro-benchmark of the BatchNorm passes that have no fused producer epilogue (EfficientNet-Lite3 backbone maps, SML B = 16, 256 x 512):
statistics, apply (+ReLU6), backward reduce + apply, through the C ABI, against the HBM time of the tensors each pass must move."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riders_amd import engine
from riders_amd.engine import _p, L, _stream
dev = torch.device("cuda:0")
lib = L()
B = int(os.environ.get("BD_B", "16"))
dt, tdt, es = 1, torch.bfloat16, 2
if os.environ.get("BD_PADDED"):      # the maps the SML really runs at 256 x 512 (input padded to 288 x 576)
    _P = {256: 288, 128: 144, 64: 72, 32: 36, 16: 18, 8: 9}
SHAPES = [(32, 128, 256, 1), (144, 128, 256, 1), (144, 64, 128, 1), (192, 64, 128, 4), (192, 32, 64, 1), (288, 32, 64, 5), (288, 16, 32, 1),
          (576, 16, 32, 10), (816, 16, 32, 9), (816, 8, 16, 1), (1392, 8, 16, 12), (32, 64, 128, 3), (48, 32, 64, 3), (96, 16, 32, 5),
          (136, 16, 32, 5), (232, 8, 16, 6)]      # (C, H, W, multiplicity): expanded maps (BN + ReLU6) then the projection outputs (BN only)
_a = torch.randn(4096, 4096, device=dev)
for _ in range(40):
    _a = (_a @ _a).clamp_(-1, 1)
torch.cuda.synchronize()


def timeit(fn, iters=50):
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


tot = [0.0] * 5
for C, H, W, mult in SHAPES:
    if os.environ.get("BD_PADDED"):
        H, W = _P[H], _P[W]
    pixels = B * H * W
    y = torch.randn((pixels, C), device=dev).to(tdt)
    z = torch.empty_like(y); dz = torch.randn((pixels, C), device=dev).to(tdt); dy = torch.empty_like(y)
    rows = lib.rd_dw_rows(pixels, C)
    stats = torch.empty((rows, C, 2), dtype=torch.float32, device=dev)
    coef = torch.empty((4, C), dtype=torch.float32, device=dev)
    gam, bet, rm, rv = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
    st = _stream(y)
    f = ctypes.c_float
    t_stats = timeit(lambda: lib.rd_bn_stats(_p(y), _p(stats), pixels, C, dt, st))
    t_fin = timeit(lambda: lib.rd_bn_finalize(_p(stats), rows, C, float(pixels), _p(gam), _p(bet), 1e-3, 0.01, 1, _p(rm), _p(rv), _p(coef[2]), _p(coef[3]), _p(coef[0]), _p(coef[1]), st))
    t_app = timeit(lambda: lib.rd_affine_act(_p(y), _p(coef[0]), _p(coef[1]), None, _p(z), pixels, C, 3, 0.0, dt, st))
    brow = lib.rd_bn_bwd_rows(pixels, C)
    part = torch.empty((brow, C, 2), dtype=torch.float32, device=dev)
    coef2 = torch.empty((2, C), dtype=torch.float32, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    t_bwd = timeit(lambda: lib.rd_bn_act_bwd_recompute(_p(dz), _p(z), _p(y), _p(coef[2]), _p(coef[3]), _p(coef[0]), _p(coef[1]), _p(part), _p(coef2), _p(dg), _p(db), 0,
                                                        _p(dy), None, pixels, C, 3, 0.0, dt, st))
    if lib.rd_bn_slab_ok(pixels, C, dt):      # the one-launch forms (rd_bn_slab.hip) next to finalize + apply / the three backward launches
        t_sf = timeit(lambda: lib.rd_bn_finalize_apply(_p(stats), rows, _p(y), _p(gam), _p(bet), 1e-3, 0.01, _p(rm), _p(rv), _p(coef[2]), _p(coef[3]), _p(coef[0]), _p(coef[1]), _p(z),
                                                       pixels, C, 3, 0.0, dt, st))
        t_sb = timeit(lambda: lib.rd_bn_act_bwd_slab(_p(dz), _p(y), _p(coef[2]), _p(coef[3]), _p(coef[0]), _p(coef[1]), _p(dg), _p(db), 0, _p(dy), pixels, C, 3, 0.0, dt, st))
        print("        one launch: finalize+apply %6.1f (was %5.1f)  backward %6.1f (was %5.1f) us" % (t_sf, t_fin + t_app, t_sb, t_bwd), flush=True)
    one = pixels * C * es / 6.0e6
    print("C=%4d %3dx%3d x%-2d stats %6.1f (%.1f)  finalize %5.1f  apply %6.1f (%.1f)  bwd %6.1f (%.1f) us" % (C, H, W, mult, t_stats, one, t_fin, t_app, 2 * one, t_bwd, 5 * one), flush=True)
    for i, v in enumerate((t_stats, t_fin, t_app, t_bwd, 8 * one)):
        tot[i] += v * mult
print("per step: stats %.0f finalize %.0f apply %.0f bwd %.0f us; HBM time of all passes %.0f us" % tuple(tot))
