"""Scale-Map-Learner parity cases (see tests/parity_cases.py for how these are used by the emulator and GPU suites)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import sml as OS
from oracle import effnet_lite3_torch as OE
from tests.golden.fill import fill_state_dict, rand_array
from tests.parity_cases import TOL, bf16_mode, close, close_l2, load, q, t


def _nhwc(x, dev, dtype=torch.float32):
    return x.to(dev).permute(0, 2, 3, 1).contiguous().to(dtype)


def _run_tape(dev, inputs, fn, gout):
    """Run fn(*nhwc inputs) under a fresh tape (tensors in the engine's activation dtype), seed grad gout (NCHW cpu) and return
    (out NCHW, grads NCHW)."""
    from riders_amd import engine
    dt = engine.act_dtype()
    xs = [_nhwc(x, dev, dt) for x in inputs]
    tape = engine.Tape()
    for x in xs:
        tape.mark(x)
    with engine._active(tape):
        out = fn(*xs)
        tape.grads[id(out)] = _nhwc(gout, dev, out.dtype)
        tape.backward()
    return out.permute(0, 3, 1, 2), [tape.grads[id(x)].permute(0, 3, 1, 2) for x in xs], tape


def bf16_exact_dwconv_case(dev, C=48, k=5, s=1, H=11, W=14, N=2, dtype=torch.bfloat16):
    """bf16 depthwise path without rounding: inputs / weights / upstream gradients in {-1,0,1} make every product and partial sum exactly
    representable, so the bf16 depthwise forward, data gradient and weight gradient must equal the fp32 torch result bit for bit
    (EfficientNet-Lite3 blocks of the SML backbone, modules/midas/blocks.py:44-64; TF-SAME padding as the blocks use it)."""
    from riders_amd import engine
    from riders_amd.midas.efficientnet_lite3 import same_pad
    rs = np.random.RandomState(C * 7 + k * 3 + s)
    w = torch.nn.Parameter(t(rs.randint(-1, 2, (C, 1, k, k)).astype(np.float32), dev))
    x = t(rs.randint(-1, 2, (N, C, H, W)).astype(np.float32))
    ph, oh = same_pad(H, k, s); pw, ow = same_pad(W, k, s)
    assert ph == pw
    tot_h, tot_w = max((oh - 1) * s + k - H, 0), max((ow - 1) * s + k - W, 0)
    xr = x.clone().requires_grad_()
    wr = w.detach().cpu().clone().requires_grad_()
    ref = F.conv2d(F.pad(xr, (pw, tot_w - pw, ph, tot_h - ph)), wr, None, stride=s, groups=C)
    assert tuple(ref.shape[2:]) == (oh, ow)
    gy = t(rs.randint(-1, 2, tuple(ref.shape)).astype(np.float32))
    (ref * gy).sum().backward()
    a = _nhwc(x, dev, dtype)
    tape = engine.Tape(); tape.mark(a)
    with engine._active(tape):
        out = engine.dwconv_block(a, w, stride=s, pad=ph, out_hw=(oh, ow))
        tape.grads[id(out)] = _nhwc(gy, dev, dtype)
        tape.backward()
    what = "bf16 dwconv C=%d k=%d s=%d" % (C, k, s)
    assert torch.equal(out.float().permute(0, 3, 1, 2).cpu(), ref.detach()), what + ": forward"
    assert torch.equal(tape.grads[id(a)].float().permute(0, 3, 1, 2).cpu(), xr.grad), what + ": data gradient"
    assert torch.equal(tape.pgrads[id(w)].cpu(), wr.grad), what + ": weight gradient"


def effnet_block_case(dev, kind="ir", cin=24, cout=32, k=3, s=2, H=12, W=16, tol=TOL, l2=False, N=2):
    """l2=True (bf16 mode): relative L2 instead of max-norm -- under bf16 rounding single activations cross a ReLU6 threshold and flip their
    mask, which a max-norm reports as a 100 % error of that element."""
    from riders_amd.midas import efficientnet_lite3 as E
    cmp = close_l2 if l2 else close
    mine = (E.InvertedResidual if kind == "ir" else E.DepthwiseSeparableConv)(cin, cout, k, s).to(dev)
    ref = (OE.InvertedResidual if kind == "ir" else OE.DepthwiseSeparableConv)(cin, cout, k, s)
    tag = "eff.%s.%d.%d.%d.%d" % (kind, cin, cout, k, s)
    ref.load_state_dict({kk: v.cpu() for kk, v in fill_state_dict(mine, tag).items()})
    x = q(t(rand_array(tag + ".x", (N, cin, H, W), 1.0)))
    xr = x.clone().requires_grad_()
    ref.train(); mine.train()
    yr = ref(xr)
    w = q(t(rand_array(tag + ".w", yr.shape, 1.0)))
    (yr * w).sum().backward()
    out, (dx,), tape = _run_tape(dev, [x], lambda a: mine._fwd(a), w)
    cmp(out, yr, tol, tag + " fwd")
    cmp(dx, xr.grad, (3 * tol if l2 else tol), tag + " dx")
    for (kk, p), (_, pr) in zip(mine.named_parameters(), ref.named_parameters()):
        cmp(tape.pgrads[id(p)], pr.grad, (3 if l2 else 2) * tol, tag + " grad " + kk)
    for kk in ("bn1.running_mean", "bn2.running_var"):
        cmp(mine.state_dict()[kk], ref.state_dict()[kk], tol, kk)


def bilinear_case(dev, tol=1e-5):
    from riders_amd import engine
    x = t(rand_array("bil.x", (2, 8, 5, 7), 1.0))
    for align in (True, False):
        xr = x.clone().requires_grad_()
        yr = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=align)
        w = t(rand_array("bil.w", yr.shape, 1.0))
        (yr * w).sum().backward()
        out, (dx,), _ = _run_tape(dev, [x], lambda a: engine.bilinear2x(a, align), w)
        close(out, yr, tol, "bilinear fwd align=%s" % align)
        close(dx, xr.grad, tol, "bilinear bwd align=%s" % align)


def fusion_block_case(dev, f=32, tol=TOL):
    from riders_amd.midas.blocks import FeatureFusionBlock_custom
    mine = FeatureFusionBlock_custom(f, torch.nn.ReLU(False), deconv=False, bn=False, expand=True, align_corners=True).to(dev)
    ref = OS.FFB(f, True)
    ref.load_state_dict({k: v.cpu() for k, v in fill_state_dict(mine, "ffb").items()})
    a, b = q(t(rand_array("ffb.a", (2, f, 6, 5), 1.0))), q(t(rand_array("ffb.b", (2, f, 6, 5), 1.0)))
    ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
    yr = ref(ar, br)
    w = q(t(rand_array("ffb.w", yr.shape, 1.0)))
    (yr * w).sum().backward()
    out, (da, db), tape = _run_tape(dev, [a, b], lambda p, q: mine._fwd(p, q), w)
    close(out, yr, tol, "ffb fwd"); close(da, ar.grad, tol, "ffb da"); close(db, br.grad, tol, "ffb db")
    for (kk, p), (_, pr) in zip(mine.named_parameters(), ref.named_parameters()):
        close(tape.pgrads[id(p)], pr.grad, 2 * tol, "ffb grad " + kk)


def loss_case(dev, tol=1e-4):
    """HIP loss vs the REFERENCE's own loss values / gradient (fixture g7) and outlier removal."""
    from riders_amd.loss import compute_loss
    from riders_amd.net_utils import OutlierRemoval
    N, H, W = 2, 24, 32
    image = t(rand_array("g7.img", (N, 1, H, W), 20.0, lo=0.05), dev)
    gi = rand_array("g7.gi", (N, 1, H, W), 30.0, lo=0.0); gi[rand_array("g7.gim", gi.shape, 1.0, lo=0.0) < 0.3] = 0
    gs = rand_array("g7.gs", (N, 1, H, W), 30.0, lo=0.0); gs[rand_array("g7.gsm", gs.shape, 1.0, lo=0.0) < 0.9] = 0
    for fs, wl in ((7, 1.5), (3, 0.0)):
        g = load("g7_loss_fs%d" % fs)
        pred = t(rand_array("g7.pred", (N, 1, H, W), 20.0, lo=0.05), dev).requires_grad_()
        loss, info = compute_loss(image=image, output_depth=pred, gt_interp=t(gi, dev), gt_sparse=t(gs, dev), loss_func='l1', w_smoothness=0.2,
                                  sobel_filter_size=fs, validity_map_loss_smoothness=torch.ones_like(image), w_lidar_loss=wl, w_edge=0.0,
                                  invalid_map_gt=None, w_unsupervised=0.0)
        got = [float(info[k]) for k in ('loss', 'loss_supervised', 'loss_lidar', 'loss_smoothness', 'loss_edge')]
        for a, b, nm in zip(got, g["loss"], ('loss', 'sup', 'lidar', 'smooth', 'edge')):
            assert abs(a - float(b)) <= tol * max(abs(float(b)), 1e-3), (fs, nm, a, float(b))
        loss.backward()
        close(pred.grad, g["dpred"], 10 * tol, "sml loss dpred fs=%d" % fs)
    # round 6: loss_func 'l2' / 'smoothl1' and w_edge > 0 against the reference's own values and gradients (fixtures g7_loss_l2 / _smoothl1 / _edge / _smoothl1_edge)
    near = (np.where(gi > 0, gi, np.where(gs > 0, gs, 10.0)) + rand_array("g7.near", (N, 1, H, W), 2.0)).astype(np.float32)
    for tag, lf, we in (("l2", "l2", 0.0), ("smoothl1", "smoothl1", 0.0), ("edge", "l1", 0.35), ("smoothl1_edge", "smoothl1", 0.5)):
        g = load("g7_loss_" + tag)
        pred = t(near, dev).requires_grad_()
        loss, info = compute_loss(image=image, output_depth=pred, gt_interp=t(gi, dev), gt_sparse=t(gs, dev), loss_func=lf, w_smoothness=0.2,
                                  sobel_filter_size=5, validity_map_loss_smoothness=torch.ones_like(image), w_lidar_loss=1.5, w_edge=we,
                                  invalid_map_gt=None, w_unsupervised=0.0)
        got = [float(info[k]) for k in ('loss', 'loss_supervised', 'loss_lidar', 'loss_smoothness', 'loss_edge')]
        for a, b, nm in zip(got, g["loss"], ('loss', 'sup', 'lidar', 'smooth', 'edge')):
            assert abs(a - float(b)) <= tol * max(abs(float(b)), 1e-3), (tag, nm, a, float(b))
        loss.backward()
        close(pred.grad, g["dpred"], 10 * tol, "sml loss dpred " + tag)
    # empty masks: the reference takes l1_loss over ZERO selected elements (utils/loss.py:56-63) -- a NaN term, a NaN total, and no gradient from that
    # term (the smoothness gradient still flows): same here, with and without any ground truth
    for gi_, gs_ in ((gi, np.zeros_like(gs)), (np.zeros_like(gi), np.zeros_like(gs))):
        pred = t(rand_array("g7.pred", (N, 1, H, W), 20.0, lo=0.05), dev).requires_grad_()
        loss, info = compute_loss(image=image, output_depth=pred, gt_interp=t(gi_, dev), gt_sparse=t(gs_, dev), loss_func='l1', w_smoothness=0.2,
                                  sobel_filter_size=7, validity_map_loss_smoothness=torch.ones_like(image), w_lidar_loss=1.5, w_edge=0.0,
                                  invalid_map_gt=None, w_unsupervised=0.0)
        loss.backward()
        pr = t(rand_array("g7.pred", (N, 1, H, W), 20.0, lo=0.05)).requires_grad_()
        lo, io = OS.compute_loss(image.cpu(), pr, t(gi_), t(gs_), 0.2, 7, torch.ones_like(image).cpu(), 1.5, 0.0, 'l1')
        lo.backward()
        assert bool(torch.isnan(lo)) and bool(torch.isnan(loss.detach().cpu())), "total with an empty mask"
        for k in ('loss_supervised', 'loss_lidar', 'loss_smoothness'):
            a, b = float(info[k]), float(io[k])
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= tol * max(abs(b), 1e-3), (k, a, b)
        assert not bool(torch.isnan(pred.grad).any()) and not bool(torch.isnan(pr.grad).any())
        close(pred.grad, pr.grad, 10 * tol, "sml loss dpred with an empty mask")
    g = load("g7_outlier")
    gt = rand_array("g7.or", (N, 1, H, W), 40.0, lo=0.0); gt[rand_array("g7.orm", gt.shape, 1.0, lo=0.0) < 0.5] = 0
    assert np.array_equal(OutlierRemoval(3, 1.5).remove_outliers(t(gt, dev)).cpu().numpy(), g["out"])
    assert np.array_equal(OutlierRemoval(7, 1.5).remove_outliers(t(gt, dev)).cpu().numpy(), g["out7"])


def prestep_case(dev):
    """Device pre-step vs the REFERENCE's scipy scales (fixture g8, |ds| <= 2e-5 = 2 x scipy's xatol) and vs the oracle's per-sample
    restatement of train_zju.py:246-343 for the assembled network inputs."""
    from riders_amd import sml_main
    g = load("g8_scale")
    H, W = 48, 64
    monos, sparses = [], []
    for i, dens in enumerate((0.02, 0.2, 0.0, 0.0005)):
        mono = rand_array("g8.mono%d" % i, (H, W), 3.0, lo=0.15)
        true_s = 0.02 + 0.05 * i
        depth = 1.0 / (true_s * mono * (1 + 0.1 * rand_array("g8.n%d" % i, (H, W), 1.0)))
        m = rand_array("g8.m%d" % i, (H, W), 1.0, lo=0.0) < dens
        monos.append(mono); sparses.append(np.where(m, depth, 0).astype(np.float32))
    B = len(monos)
    mono = t(np.stack(monos)[:, None], dev).contiguous(); sparse = t(np.stack(sparses)[:, None], dev).contiguous()
    image = t(rand_array("g8.img", (B, 3, H, W), 1.0, lo=0.0), dev)
    rc = rand_array("g8.rc", (B, 1, H, W), 50.0, lo=0.02); rc[rand_array("g8.rcm", rc.shape, 1.0, lo=0.0) < 0.9] = 0
    hw = (32, 32)
    x, d, scale = sml_main.prepare_inputs(image, mono, sparse, t(rc, dev), hw)
    sc = scale.cpu().numpy()
    for i in range(B):
        assert abs(float(sc[i]) - float(g["s%d" % i][0])) <= 2e-5, (i, float(sc[i]), float(g["s%d" % i][0]))
        xo, do, _ = OS.prestep_sample(image[i].cpu().numpy(), monos[i], sparses[i], rc[i, 0], hw, scale=float(sc[i]))
        close(d[i], do, 1e-5, "prestep d[%d]" % i)
        close(x[i], xo, 1e-4, "prestep x[%d]" % i)
        close((monos[i] * np.float32(sc[i])).clip(1 / 255.0, 10.0)[::4, ::4], g["out%d" % i], 5e-4, "int_depth vs reference")


def prestep_st_case(dev):
    """'st' global alignment on device (rd_sml_scale_shift_ls, train_zju.py:278-287) vs the REFERENCE's LeastSquaresEstimator values
    (fixture g8 `ls*`, 1e-3 relative as north_star; the singular sample gives exactly (0, 0)) and vs the oracle's per-sample
    restatement for the assembled inputs."""
    from riders_amd import sml_main
    g = load("g8_scale")
    H, W = 48, 64
    monos, sparses = [], []
    for i, dens in enumerate((0.02, 0.2, 0.0, 0.0005)):
        mono = rand_array("g8.mono%d" % i, (H, W), 3.0, lo=0.15)
        true_s = 0.02 + 0.05 * i
        depth = 1.0 / (true_s * mono * (1 + 0.1 * rand_array("g8.n%d" % i, (H, W), 1.0)))
        m = rand_array("g8.m%d" % i, (H, W), 1.0, lo=0.0) < dens
        monos.append(mono); sparses.append(np.where(m, depth, 0).astype(np.float32))
    B = len(monos)
    mono = t(np.stack(monos)[:, None], dev).contiguous(); sparse = t(np.stack(sparses)[:, None], dev).contiguous()
    image = t(rand_array("g8.img", (B, 3, H, W), 1.0, lo=0.0), dev)
    rc = rand_array("g8.rc", (B, 1, H, W), 50.0, lo=0.02); rc[rand_array("g8.rcm", rc.shape, 1.0, lo=0.0) < 0.9] = 0
    hw = (32, 32)
    cfg = dict(sml_main.ZJU_SML_CONFIG, global_alignment='st')
    x, d, (scale, shift) = sml_main.prepare_inputs(image, mono, sparse, t(rc, dev), hw, cfg)
    sc, sf = scale.cpu().numpy(), shift.cpu().numpy()
    for i in range(B):
        rs, rt = float(g["ls%d" % i][0]), float(g["ls%d" % i][1])
        if rs == 0.0 and rt == 0.0:
            assert float(sc[i]) == 0.0 and float(sf[i]) == 0.0, (i, sc[i], sf[i])
        else:
            assert abs(float(sc[i]) - rs) <= 1e-3 * abs(rs), (i, float(sc[i]), rs)
            # the shift is a small difference of large sums: bound it by what a 1e-3 scale error does to the fit at the data's mean
            assert abs(float(sf[i]) - rt) <= 1e-3 * (abs(rt) + abs(rs) * float(monos[i].mean())), (i, float(sf[i]), rt)
        xo, do, _ = OS.prestep_sample(image[i].cpu().numpy(), monos[i], sparses[i], rc[i, 0], hw, scale=float(sc[i]), shift=float(sf[i]),
                                      global_alignment='st')
        close(d[i], do, 1e-5, "prestep('st') d[%d]" % i)
        close(x[i], xo, 1e-4, "prestep('st') x[%d]" % i)


def metrics_case(dev):
    import ctypes
    from riders_amd import engine
    g = load("g11_metrics")
    inv = rand_array("g11.inv", (1, 1, 36, 48), 0.3, lo=0.03) + np.float32(0.01)
    gt = rand_array("g11.gt", (60, 80), 60.0, lo=0.0); gt[rand_array("g11.m", gt.shape, 1.0, lo=0.0) < 0.9] = 0
    lib, p = engine.L(), engine._p
    depth = (1.0 / t(inv)).to(dev).contiguous()
    up = torch.empty((1, 1, 60, 80), dtype=torch.float32, device=dev)
    assert lib.rd_bicubic_resize(p(depth), p(up), 1, 36, 48, 60, 80, engine._stream(depth)) == 0
    close(up[0, 0], g["pred"], 1e-5, "bicubic")
    res = torch.empty((1, 8), dtype=torch.float64, device=dev)
    gtd = t(gt, dev).contiguous()
    assert lib.rd_depth_metrics(p(up), p(gtd), 1, 60 * 80, ctypes.c_float(0.0), ctypes.c_float(50.0), p(res), engine._stream(up)) == 0
    r = res.cpu().numpy()[0]
    got = [r[1] / r[0], np.sqrt(r[2] / r[0]), r[3] / r[0], np.sqrt(r[4] / r[0]), r[5] / r[0], r[6] / r[0], r[7] / r[0]]
    for a, b, nm in zip(got, g["vals"], ("mae", "rmse", "imae", "irmse", "abs_rel", "sq_rel", "delta1")):
        assert abs(a - b) <= 1e-3 * max(abs(b), 1e-6), (nm, a, b)   # abs-rel within 1e-3 (north_star)


def sml_net_bf16_case(dev, B=4, H=128, W=192):
    """configs[2] runs the SML in bf16 (activations stored as bf16; fp32 accumulation, parameters, statistics).  What the kernels owe is
    the arithmetic of that mode: the oracle is run with the identical rounding points (oracle/precision.py).  At layer / block level the
    HIP path reproduces it to 1e-4 .. 2e-3 relative L2 (sml bf16 block cases).  Through the whole 80-layer network two implementations
    of the same bf16 arithmetic decorrelate to rounding-noise level -- a sub-ulp difference in a pre-rounding value flips the rounding with
    proportional probability, so any difference grows to ~1 ulp within a few layers (tools/debug_sml_bf16_emu.py: 0.15 % per block in
    eval mode) -- and training-mode BatchNorm amplifies the rounding of its input by |mean| / std per layer.  Hence the network-level
    bounds are those of the number format, not of the kernels: eval-mode prediction within 2e-2 relative L2 of the emulating oracle,
    train-mode prediction within 0.15 (measured 7e-2; the plain fp32 path is 0.2 away from either, tools/probe_sml_bf16.py)."""
    from riders_amd.midas.midas_net_custom import MidasNet_small_videpth
    with bf16_mode():
        m = MidasNet_small_videpth(device=dev, min_pred=0.1, max_pred=255.0, in_channels=3)
        sd = fill_state_dict(m, "g9.sml")
        xin = rand_array("s16.x", (B, 3, H, W), 1.0)
        din = rand_array("s16.d", (B, 1, H, W), 0.3, lo=0.05) + np.float32(0.02)
        o = OS.SMLOracle()
        o.load_state_dict({k: v.cpu() for k, v in sd.items()})
        x, d = t(xin, dev), t(din, dev)
        errs = {}
        with torch.no_grad():
            m.train(); o.train()
            errs["pred train"] = close_l2(m.forward(x, d), o(t(xin), t(din)), 0.15, "SML train-mode pred (bf16 vs bf16-emulating oracle)")
            o.load_state_dict({k: v.cpu() for k, v in sd.items()}); fill_state_dict(m, "g9.sml")      # undo the running-statistics update
            m.eval(); o.eval()
            errs["pred eval"] = close_l2(m.forward(x, d), o(t(xin), t(din)), 2e-2, "SML eval-mode pred (bf16 vs bf16-emulating oracle)")
        print("SML bf16 HIP vs bf16-emulating oracle (B=%d %dx%d): " % (B, H, W) + "  ".join("%s %.2e" % kv for kv in errs.items()))


def sml_config2_fullsize_case(dev):
    """configs[2] at ITS OWN size (B = 16, 256x512 frames, bf16): the launches the throughput figure runs on -- kernel routing depends on
    pixel counts (conv_few_kernel, persistent conv1x1_direct, rd_conv_fwd_streams, depthwise block geometry, frag conv tiles).
    (a) fp32, B = 2 at 256x512: train-mode prediction and loss against the oracle within 1e-3;
    (b) bf16, B = 16: a full training step is finite and bit-reproducible (two runs from the same state give identical loss and
        parameters), and the eval-mode prediction of a 4-sample subset agrees with the bf16-emulating oracle within 2e-2 relative L2."""
    from riders_amd import engine, sml_main
    from riders_amd.midas.midas_net_custom import MidasNet_small_videpth
    from riders_amd.optim import FlatAdam
    H, W = 256, 512
    # (a) fp32 vs oracle
    m = MidasNet_small_videpth(device=dev, min_pred=0.1, max_pred=255.0, in_channels=3)
    sd = fill_state_dict(m, "g9.sml")
    xin = rand_array("s17.x", (2, 3, H, W), 1.0)
    din = rand_array("s17.d", (2, 1, H, W), 0.3, lo=0.05) + np.float32(0.02)
    o = OS.SMLOracle(); o.load_state_dict({k: v.cpu() for k, v in sd.items()})
    m.train(); o.train()
    with torch.no_grad():
        close(m.forward(t(xin, dev), t(din, dev)), o(t(xin), t(din)), TOL, "SML 256x512 fp32 train-mode pred")
    # (b) bf16 B = 16
    with bf16_mode():
        finals = []
        batch = sml_main.synthetic_batch(16, H, W, seed=31, device=dev)
        for rep in range(2):
            torch.manual_seed(0)
            m = MidasNet_small_videpth(device=dev, min_pred=0.1, max_pred=255.0, in_channels=3)
            fill_state_dict(m, "g9.sml")
            m.train()
            opt = FlatAdam(m.parameters(), lr=1e-4)
            orr = sml_main.make_outlier_removal()
            losses = [float(sml_main.train_step(m, opt, batch, outlier=orr)) for _ in range(2)]
            assert all(np.isfinite(losses)), losses
            finals.append((losses, opt.flat_param.clone()))
        assert finals[0][0] == finals[1][0] and torch.equal(finals[0][1], finals[1][1]), "SML bf16 B=16 step is not reproducible"
        m = MidasNet_small_videpth(device=dev, min_pred=0.1, max_pred=255.0, in_channels=3)
        sd = fill_state_dict(m, "g9.sml")
        m.eval()
        o = OS.SMLOracle(); o.load_state_dict({k: v.cpu() for k, v in sd.items()}); o.eval()
        xin = rand_array("s18.x", (16, 3, H, W), 1.0)
        din = rand_array("s18.d", (16, 1, H, W), 0.3, lo=0.05) + np.float32(0.02)
        with torch.no_grad():
            pred = m.forward(t(xin, dev), t(din, dev))
            ref = o(t(xin[[0, 5, 10, 15]]), t(din[[0, 5, 10, 15]]))
        close_l2(pred[[0, 5, 10, 15]], ref, 2e-2, "SML B=16 256x512 bf16 eval-mode pred (sample subset) vs bf16-emulating oracle")


def sml_bf16_convergence_case(dev, steps=150):
    """does bf16 TRAIN like fp32?  150 optimisation steps of the SML on one synthetic batch (B = 4, 128x192) from
    identical weights in both precisions, plus an fp32 run whose initial weights are perturbed by 1e-6 relative -- the yardstick for what
    "the same training run" means in fp32 itself (tools/probe_sml_convergence.py: the synthetic task falls from 126 to a plateau at 65-67
    within ten steps, with loss excursions to ~530 around step 33).  The run is chaotic at the level of single steps: over round 3's
    kernel changes (a summation order in the BatchNorm backward, fused multiply-adds in the depthwise taps -- each a last-bit change) the
    fp32 curve itself moved by up to 2.3 % between builds and its perturbed copy by as much, the bf16 first loss between 129.4 and
    131.4 (fp32: 126.5), the excursion count between 1 and 3 and its height between 215 and 560.  Stated bands, tied to that yardstick:
    first loss within 5 %, both runs fall below 0.7x their first loss, the 10-step window medians from step 10 within max(3 %, 3x the
    fp32 self-sensitivity) of the fp32 curve, excursion heights within a factor of 5 where both runs have one, final eval-mode abs-rel
    within 10 % relative.  (The per-step training-mode PREDICTIONS differ by ~0.2 relative L2 -- bf16 rounding amplified by batch
    statistics -- without moving the loss.)"""
    from riders_amd import engine, sml_main
    from riders_amd.midas.midas_net_custom import MidasNet_small_videpth
    from riders_amd.optim import FlatAdam
    batch_cpu = sml_main.synthetic_batch(4, 128, 192, seed=41)
    curves, absrel = {}, {}
    for mode in ("fp32", "bf16", "fp32 perturbed"):
        engine.set_compute_dtype(mode[:4]); engine.clear_caches()
        try:
            torch.manual_seed(0)
            m = MidasNet_small_videpth(device=dev, min_pred=0.1, max_pred=255.0, in_channels=3)
            fill_state_dict(m, "g9.sml")
            m.train()
            opt = FlatAdam(m.parameters(), lr=2e-4)
            if mode.endswith("perturbed"):
                with torch.no_grad():
                    opt.flat_param.mul_(1.0 + 1e-6)
                engine.refresh_packed()
            orr = sml_main.make_outlier_removal()
            batch = tuple(b.to(dev) for b in batch_cpu)
            curves[mode] = np.array([float(sml_main.train_step(m, opt, batch, outlier=orr)) for _ in range(steps)])
            m.eval()
            absrel[mode] = float(np.mean(sml_main.validate_batch(m, batch)["abs_rel"]))
        finally:
            engine.set_compute_dtype("fp32"); engine.clear_caches()
    f, b, p = curves["fp32"], curves["bf16"], curves["fp32 perturbed"]
    assert np.all(np.isfinite(b)) and np.all(np.isfinite(f))
    # the task has one loss excursion around step 33 (fp32: 62 -> 534 -> 64); bf16 meets it three steps later, the perturbed fp32 run at
    # the same step with a 2 % different height: curves are compared through 10-step window medians, excursions by count and height
    def medians(c):
        return np.array([np.median(c[i:i + 10]) for i in range(10, steps - 9, 10)])

    def spikes(c):
        med = np.median(c[10:])
        return c[10:][c[10:] > 2.0 * med]
    mf, mb, mp = medians(f), medians(b), medians(p)
    dev_b, dev_p = np.abs(mb - mf) / mf, np.abs(mp - mf) / mf
    sf, sb = spikes(f), spikes(b)
    print("SML convergence: loss[0] %.3f / %.3f (fp32 / bf16), loss[-1] %.3f / %.3f, 10-step medians from step 10: max |bf16 - fp32| / fp32 %.4f, fp32 "
          "self-sensitivity (1e-6 weight perturbation) %.4f, excursions %s / %s, abs-rel %.4f / %.4f" % (
              f[0], b[0], f[-1], b[-1], dev_b.max(), dev_p.max(), np.round(sf, 1).tolist(), np.round(sb, 1).tolist(), absrel["fp32"], absrel["bf16"]))
    assert abs(b[0] - f[0]) <= 0.05 * f[0]
    assert f[-1] < 0.7 * f[0] and b[-1] < 0.7 * b[0]
    assert dev_b.max() <= max(0.03, 3.0 * dev_p.max()), (dev_b.max(), dev_p.max())
    # an excursion is ONE step of a transient: neither its height nor how often it recurs is reproducible between builds (docstring)
    assert len(sb) == 0 or len(sf) == 0 or 0.2 * sf.max() <= sb.max() <= 5.0 * sf.max(), (sf, sb)
    assert abs(absrel["bf16"] - absrel["fp32"]) <= 0.10 * absrel["fp32"], absrel


def validate_chain_case(dev, tol=1e-3):
    """H3 (val_zju.py:124-254) end to end on identical weights (g9 fill) and inputs: device pre-step -> network (eval) -> 1/pred ->
    bicubic to the frame size -> masked metrics, against the oracle chain prestep_sample -> SMLOracle -> val_metrics.
    north_star: abs-rel within 1e-3 (absolute) of the CPU path; the other metrics within 1e-3 relative."""
    from riders_amd import sml_main
    from riders_amd.midas.midas_net_custom import MidasNet_small_videpth
    m = MidasNet_small_videpth(device=dev, min_pred=0.1, max_pred=255.0, in_channels=3)
    sd = fill_state_dict(m, "g9.sml")
    m.eval()
    B, H, W = 2, 60, 80                                   # frames are resized (nearest) to the 288 x 384 network input, as for ZJU
    batch = sml_main.synthetic_batch(B, H, W, seed=21)
    got = sml_main.validate_batch(m, tuple(b.to(dev) for b in batch))
    o = OS.SMLOracle()
    o.load_state_dict({k: v.cpu() for k, v in sd.items()})
    o.eval()
    image, mono, radar, gt, sparse_gt, rcnet = [b.numpy() for b in batch]
    hw = sml_main.net_size(H, W)
    for i in range(B):
        xo, do, _ = OS.prestep_sample(image[i], mono[i, 0], radar[i, 0], rcnet[i, 0], hw)
        with torch.no_grad():
            po = o(t(xo)[None], t(do)[None])
        ref = OS.val_metrics(po, sparse_gt[i, 0], (H, W))
        assert abs(float(got["abs_rel"][i]) - float(ref["abs_rel"])) <= tol, ("abs_rel", i, got["abs_rel"][i], ref["abs_rel"])
        for k in ("mae", "rmse", "imae", "irmse", "sq_rel", "delta1"):
            assert abs(float(got[k][i]) - float(ref[k])) <= tol * max(abs(float(ref[k])), 1e-6), (k, i, got[k][i], ref[k])
        close(got["depth"][i, 0], ref["pred"], tol, "validation depth map %d" % i)


def sml_net_case(dev, tol=TOL):
    """Full MidasNet_small_videpth forward/backward vs the REFERENCE's own outputs (fixture g9; hub backbone = the restated one)."""
    from riders_amd.midas.midas_net_custom import MidasNet_small_videpth
    g = load("g9_sml")
    m = MidasNet_small_videpth(device=dev, min_pred=0.1, max_pred=255.0, in_channels=3)
    fill_state_dict(m, "g9.sml")
    B, H, W = 2, 64, 96
    x = t(rand_array("g9.x", (B, 3, H, W), 1.0), dev).requires_grad_()
    d = t(rand_array("g9.d", (B, 1, H, W), 0.3, lo=0.05) + np.float32(0.02), dev)
    m.train()
    pred = m.forward(x, d)
    close(pred, g["pred"], tol, "g9 pred")
    (pred * t(rand_array("g9.w", pred.shape, 1.0), dev)).sum().backward()
    sd = m.state_dict()
    close(sd["first.1.running_mean"], g["rm_first"], tol, "first BN running mean")
    close(sd["pretrained.layer1.1.running_var"], g["rv_stem"], tol, "stem BN running var")
    # conditioning-aware gradient check against an fp64 run of the oracle (see tests/parity_cases.py)
    def oracle_grads(dt):
        o = OS.SMLOracle().to(dt)
        o.load_state_dict({k: (v.to(dt) if v.is_floating_point() else v) for k, v in fill_state_dict(MidasNet_small_videpth(
            device='cpu', min_pred=0.1, max_pred=255.0, in_channels=3), "g9.sml").items()})
        o.train()
        xo = t(rand_array("g9.x", (B, 3, H, W), 1.0)).to(dt).requires_grad_()
        po = o(xo, (t(rand_array("g9.d", (B, 1, H, W), 0.3, lo=0.05) + np.float32(0.02))).to(dt))
        (po * t(rand_array("g9.w", po.shape, 1.0)).to(dt)).sum().backward()
        gr = {k: p.grad for k, p in o.named_parameters()}
        gr["x"] = xo.grad
        return gr
    g64, g32 = oracle_grads(torch.float64), oracle_grads(torch.float32)

    def cond(k):
        return float((g32[k].double() - g64[k]).abs().max() / max(float(g64[k].abs().max()), 1e-30))
    close(x.grad, g64["x"], max(4 * tol, 3 * cond("x")), "g9 dx")
    close(g32["x"], g["dx"], max(4 * tol, 3 * cond("x")), "oracle vs reference dx")
    # B=2 at 64x96 leaves 12 samples per BatchNorm channel in layer4: single parameters are ill-conditioned in fp32 (one fp32 oracle run is
    # itself a noisy estimate of that), so the per-parameter bound is loose (50 x) and the decisive check is the global relative L2 error of
    # ALL parameter gradients, which must stay within 3 x the fp32 oracle's own global error (floor 2e-3).  Measured: with the BatchNorm
    # statistics summed in the depthwise epilogue's order instead of rd_bn_stats' order (both exact to 7e-8, the block itself matches the
    # oracle to 1e-6 either way) the worst parameter -- layer4.1.0.bn1.bias, upstream of a 1392-channel BatchNorm over a 2x3 map -- moves
    # from 2 x to 34 x its conditioning estimate: rounding-level differences amplified ~3e4 x by that layer, not a kernel difference.
    num = den = num32 = 0.0
    for k, p in m.named_parameters():
        if (k + "|none") in g:
            assert p.grad is None, k
            continue
        ref = g64[k]
        num += float((p.grad.detach().cpu().double() - ref).pow(2).sum())
        num32 += float((g32[k].double() - ref).pow(2).sum())
        den += float(ref.pow(2).sum())
        if float(ref.abs().max()) < 1e-12:
            continue
        close(p.grad, ref, max(4 * tol, 50 * cond(k)), "g9 grad " + k)
    gerr, gcond = (num / den) ** 0.5, (num32 / den) ** 0.5
    assert gerr <= max(2e-3, 3 * gcond), "global gradient error %.3e vs fp32-oracle %.3e" % (gerr, gcond)
    m.eval()
    with torch.no_grad():
        close(m.forward(x.detach(), d), g["pred_eval"], tol, "g9 eval pred")


def sml_fullsize_backward_case(dev, tol=TOL):
    """the SML BACKWARD at configs[2]'s frame size (fp32, B = 2, 256x512), i.e. at the pixel counts that select
    conv_few_kernel, the persistent conv1x1_direct, the depthwise block geometry and the frag-conv tiles of the throughput run.  Input
    gradient and the GLOBAL relative L2 error of all parameter gradients against an fp64 run of the oracle, with the conditioning-aware bound
    of sml_net_case (3 x the fp32 oracle's own global error, floor 2e-3); the prediction itself within 1e-3."""
    from riders_amd.midas.midas_net_custom import MidasNet_small_videpth
    B, H, W = 2, 256, 512
    m = MidasNet_small_videpth(device=dev, min_pred=0.1, max_pred=255.0, in_channels=3)
    fill_state_dict(m, "g9.sml")
    xin = rand_array("s19.x", (B, 3, H, W), 1.0)
    din = rand_array("s19.d", (B, 1, H, W), 0.3, lo=0.05) + np.float32(0.02)
    win = rand_array("s19.w", (B, 1, H, W), 1.0)
    x = t(xin, dev).requires_grad_()
    m.train()
    pred = m.forward(x, t(din, dev))
    (pred * t(win, dev)).sum().backward()

    def oracle_grads(dt):
        o = OS.SMLOracle().to(dt)
        o.load_state_dict({k: (v.to(dt) if v.is_floating_point() else v) for k, v in fill_state_dict(MidasNet_small_videpth(
            device='cpu', min_pred=0.1, max_pred=255.0, in_channels=3), "g9.sml").items()})
        o.train()
        xo = t(xin).to(dt).requires_grad_()
        po = o(xo, t(din).to(dt))
        (po * t(win).to(dt)).sum().backward()
        gr = {k: p.grad for k, p in o.named_parameters()}
        gr["x"] = xo.grad
        return po.detach(), gr
    p64, g64 = oracle_grads(torch.float64)
    p32, g32 = oracle_grads(torch.float32)
    close(pred, p64, tol, "SML 256x512 fp32 train-mode pred")

    def gl2(get):
        num = den = 0.0
        for k, ref in g64.items():
            if ref is None:
                continue
            v = get(k)
            assert v is not None, k
            num += float((v.detach().cpu().double() - ref).pow(2).sum()); den += float(ref.pow(2).sum())
        return (num / den) ** 0.5
    hip = {k: p.grad for k, p in m.named_parameters()}
    hip["x"] = x.grad
    for k, ref in g64.items():
        if ref is None:
            assert hip.get(k) is None, "unexpected gradient for " + k
    gerr, gcond = gl2(lambda k: hip[k]), gl2(lambda k: g32[k])
    print("SML 256x512 backward: global gradient error %.3e (fp32 oracle vs fp64: %.3e)" % (gerr, gcond))
    assert gerr <= max(2e-3, 3 * gcond), "global gradient error %.3e vs fp32-oracle %.3e" % (gerr, gcond)
    cx = float((g32["x"].double() - g64["x"]).abs().max() / float(g64["x"].abs().max()))
    close(x.grad, g64["x"], max(4 * tol, 3 * cx), "SML 256x512 dx")


def sml_config3_rank_case(dev, tol=TOL):
    """The SML's per-rank share of configs[3] (global batch 32 on 8 GPUs = B = 4 per rank, 256x512 frames): fp32 loss of the whole step's
    forward (device pre-step -> network -> 1/pred -> outlier removal -> loss) against the oracle chain within 1e-3 stage by stage, then the
    bf16 graphed training step at that size is finite and bit-reproducible."""
    from riders_amd import engine, sml_main
    from riders_amd.optim import FlatAdam
    cfg = sml_main.ZJU_SML_CONFIG
    B, H, W = 4, 256, 512
    batch = sml_main.synthetic_batch(B, H, W, seed=33)
    torch.manual_seed(0)
    m = sml_main.build_model(dev, cfg)
    fill_state_dict(m, "g9.sml")
    m.train()
    o = OS.SMLOracle(); o.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()}); o.train()
    with torch.no_grad():
        dbatch = tuple(b.to(dev) for b in batch)
        image, mono, radar, gt, sparse_gt, rcnet = [b.numpy() for b in batch]
        hw = sml_main.net_size(H, W)
        xs, ds = [], []
        for i in range(B):
            xo, do, _ = OS.prestep_sample(image[i], mono[i, 0], radar[i, 0], rcnet[i, 0], hw)
            xs.append(t(np.ascontiguousarray(xo))); ds.append(t(np.ascontiguousarray(do)))
        xo, do = torch.stack(xs).float(), torch.stack(ds).float()
        # stage by stage: device pre-step, network, then the loss chain evaluated on the HIP prediction (the loss is dominated by the far
        # depths = the smallest predictions, where a prediction error of 1e-3 of max|pred| is a much larger relative error of 1/pred: the
        # end-to-end loss of two 1e-3-close predictions differs by 3e-3 here, so the loss kernels are checked on identical predictions)
        x, d, _ = sml_main.prepare_inputs(dbatch[0], dbatch[1], dbatch[2], dbatch[5], hw, cfg)
        close(x, xo, tol, "SML B=4 pre-step network input"); close(d, do, tol, "SML B=4 pre-step scaffold")
        # the network on IDENTICAL inputs (the oracle pre-step's): train-mode BatchNorm over four frames whose scaffolds are piecewise
        # constant is badly conditioned with respect to its INPUT (the two pre-steps agree to 1e-6 in the scale factor -- Brent vs the exact
        # weighted median -- and the predictions on the two inputs differ by far more than the kernels do; printed below)
        ref_pred = o(xo, do)
        close(m.forward(xo.to(dev), do.to(dev)), ref_pred, tol, "SML B=4 256x512 fp32 train-mode pred (identical inputs)")
        pred = m.forward(x, d)
        sens = float((pred.float().cpu() - ref_pred).abs().max() / ref_pred.abs().max())
        print("SML B=4: prediction on the device pre-step's inputs vs on the oracle pre-step's: max-norm difference %.2e" % sens)
        loss = float(sml_main.forward_loss(m, dbatch, cfg, sml_main.make_outlier_removal(cfg)))
        gi = torch.stack([t(np.ascontiguousarray(OS.nearest_resize(gt[i, 0], hw[0], hw[1]))) for i in range(B)])[:, None].float()
        gs = torch.stack([t(np.ascontiguousarray(OS.nearest_resize(sparse_gt[i, 0], hw[0], hw[1]))) for i in range(B)])[:, None].float()
        gi = OS.remove_outliers(gi, cfg['outlier_removal_kernel_size'], cfg['outlier_removal_threshold'])
        ref, _ = OS.compute_loss(1.0 / do, 1.0 / pred.float().cpu(), gi, gs, w_smoothness=cfg['w_smoothness'], sobel_filter_size=cfg['sobel_filter_size'],
                                 w_lidar_loss=cfg['w_lidar_loss'], w_edge=cfg['w_edge'])
    assert abs(loss - float(ref)) <= tol * abs(float(ref)), (loss, float(ref))
    with bf16_mode():
        runs = []
        for rep in range(2):
            torch.manual_seed(0)
            m = sml_main.build_model(dev, cfg)
            fill_state_dict(m, "g9.sml")
            m.train()
            opt = FlatAdam(m.parameters(), lr=cfg['learning_rate'])
            step = sml_main.GraphedTrainStep(m, opt, dbatch, cfg, outlier=sml_main.make_outlier_removal(cfg), warmup=1)
            ls = [float(step()) for _ in range(3)]
            assert all(np.isfinite(ls)), ls
            runs.append((ls, opt.flat_param.clone()))
            del step
            engine.set_param_grad_allocator(None)
        assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1]), "SML B=4 bf16 graphed step is not reproducible"


def sml_config4_rank_case(dev):
    """BASELINE.json configs[4], the SML half at ITS per-rank size: batch 8 of 512x1024 frames run at NATIVE resolution (cfg['net_hw'] =
    (512, 1024); the reference's transform would shrink them to 288x576 like every other frame, which is what configs[2] measures), fp16.  (a) eval-mode
    prediction of all 8 frames in fp16 against the fp32 HIP path on identical weights / inputs (relative L2 <= 1e-2; the fp32 path is pinned to
    the oracle by the other cases); (b) the loss-scaled graphed fp16 training step (device pre-step + forward + loss + backward + Adam, scale
    1024 folded back by Adam) is finite, skips nothing, and is bit-reproducible from the same state."""
    from riders_amd import engine, sml_main
    from riders_amd.optim import FlatAdam
    B, H, W = 8, 512, 1024
    cfg = dict(sml_main.ZJU_SML_CONFIG, net_hw=(H, W))
    sb = sml_main.synthetic_batch(B, H, W, seed=47, device=dev)
    hw = (H, W)
    preds = {}
    for mode in ("fp32", "fp16"):
        engine.set_compute_dtype(mode); engine.clear_caches()
        try:
            torch.manual_seed(0)
            m = sml_main.build_model(dev, cfg)
            m.eval()
            with torch.no_grad():
                x, d, _ = sml_main.prepare_inputs(sb[0], sb[1], sb[2], sb[5], hw, cfg)
                preds[mode] = m.forward(x, d).float().cpu()
            del m
        finally:
            engine.set_compute_dtype("fp32"); engine.clear_caches()
    assert tuple(preds["fp16"].shape) == (B, 1, 512, 1024)
    close_l2(preds["fp16"], preds["fp32"], 1e-2, "configs[4] SML B=8 512x1024 eval prediction fp16 vs fp32")
    engine.set_compute_dtype("fp16"); engine.clear_caches()
    try:
        runs = []
        for rep in range(2):
            torch.manual_seed(0)
            m = sml_main.build_model(dev, cfg)
            m.train()
            opt = FlatAdam(m.parameters(), lr=cfg['learning_rate'])
            step = sml_main.GraphedTrainStep(m, opt, sb, cfg, outlier=sml_main.make_outlier_removal(cfg), warmup=1, loss_scale=1024.0)
            ls = [float(step()) for _ in range(3)]
            assert all(np.isfinite(ls)) and opt.skipped_steps() == 0 and bool(torch.isfinite(opt.flat_param).all()), (ls, opt.skipped_steps())
            runs.append((ls, opt.flat_param.clone()))
            del step, opt, m
            engine.set_param_grad_allocator(None)
            engine.clear_caches()
        assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1]), "configs[4] SML B=8 fp16 graphed step is not reproducible"
    finally:
        engine.set_compute_dtype("fp32"); engine.clear_caches(); engine.set_param_grad_allocator(None)
