"""Pin the SML oracle against golden vectors produced by the reference (tests/golden/make_golden_sml.py)."""
import numpy as np
import torch

from oracle import sml as OS
from tests.golden.fill import fill_state_dict, rand_array
from tests.parity_cases import close, load, t


def test_g7_loss_and_outlier():
    N, H, W = 2, 24, 32
    image = t(rand_array("g7.img", (N, 1, H, W), 20.0, lo=0.05))
    gi = rand_array("g7.gi", (N, 1, H, W), 30.0, lo=0.0); gi[rand_array("g7.gim", gi.shape, 1.0, lo=0.0) < 0.3] = 0
    gs = rand_array("g7.gs", (N, 1, H, W), 30.0, lo=0.0); gs[rand_array("g7.gsm", gs.shape, 1.0, lo=0.0) < 0.9] = 0
    for fs, wl in ((7, 1.5), (3, 0.0)):
        g = load("g7_loss_fs%d" % fs)
        pred = t(rand_array("g7.pred", (N, 1, H, W), 20.0, lo=0.05)).requires_grad_()
        loss, info = OS.compute_loss(image, pred, t(gi), t(gs), 0.2, fs, None, wl, 0.0)
        got = [float(info[k]) for k in ('loss', 'loss_supervised', 'loss_lidar', 'loss_smoothness', 'loss_edge')]
        assert np.allclose(got, g["loss"], rtol=1e-5, atol=1e-6), (got, g["loss"])
        loss.backward()
        close(pred.grad, g["dpred"], 1e-5)
    # round 6: 'l2' / 'smoothl1' and the edge-matching term (utils/loss.py:72-100, :241-249), fixtures from the reference's compute_loss
    near = (np.where(gi > 0, gi, np.where(gs > 0, gs, 10.0)) + rand_array("g7.near", (N, 1, H, W), 2.0)).astype(np.float32)
    for tag, lf, we in (("l2", "l2", 0.0), ("smoothl1", "smoothl1", 0.0), ("edge", "l1", 0.35), ("smoothl1_edge", "smoothl1", 0.5)):
        g = load("g7_loss_" + tag)
        pred = t(near).requires_grad_()
        loss, info = OS.compute_loss(image, pred, t(gi), t(gs), 0.2, 5, None, 1.5, we, lf)
        got = [float(info[k]) for k in ('loss', 'loss_supervised', 'loss_lidar', 'loss_smoothness', 'loss_edge')]
        assert np.allclose(got, g["loss"], rtol=1e-5, atol=1e-6), (tag, got, g["loss"])
        loss.backward()
        close(pred.grad, g["dpred"], 1e-5)
    g = load("g7_outlier")
    gt = rand_array("g7.or", (N, 1, H, W), 40.0, lo=0.0); gt[rand_array("g7.orm", gt.shape, 1.0, lo=0.0) < 0.5] = 0
    assert np.array_equal(OS.remove_outliers(t(gt), 3, 1.5).numpy(), g["out"])
    assert np.array_equal(OS.remove_outliers(t(gt), 7, 1.5).numpy(), g["out7"])


def test_g8_scale():
    g = load("g8_scale")
    H, W = 48, 64
    for i, dens in enumerate((0.02, 0.2, 0.0, 0.0005)):
        mono = rand_array("g8.mono%d" % i, (H, W), 3.0, lo=0.15)
        true_s = 0.02 + 0.05 * i
        depth = 1.0 / (true_s * mono * (1 + 0.1 * rand_array("g8.n%d" % i, (H, W), 1.0)))
        m = rand_array("g8.m%d" % i, (H, W), 1.0, lo=0.0) < dens
        sparse = np.where(m, depth, 0).astype(np.float32)
        valid = (sparse < 100.0) * (sparse > 0.0)
        tgt = sparse.copy(); tgt[~valid] = np.inf; tgt = 1.0 / tgt
        assert abs(OS.optimize_scale(mono, tgt, valid) - float(g["s%d" % i][0])) < 1e-9
        # 'st' closed form (modules/estimator.py:5-29) against the reference's LeastSquaresEstimator output stored in the same fixture
        s_ls, t_ls = OS.scale_and_shift_ls(mono, tgt, valid)
        assert abs(s_ls - float(g["ls%d" % i][0])) <= 1e-6 * max(1.0, abs(float(g["ls%d" % i][0]))), (i, s_ls, g["ls%d" % i])
        assert abs(t_ls - float(g["ls%d" % i][1])) <= 1e-6, (i, t_ls, g["ls%d" % i])


def test_g9_sml_network():
    from riders_amd.midas.midas_net_custom import MidasNet_small_videpth
    g = load("g9_sml")
    o = OS.SMLOracle()
    o.load_state_dict(fill_state_dict(MidasNet_small_videpth(device='cpu', min_pred=0.1, max_pred=255.0, in_channels=3), "g9.sml"))
    B, H, W = 2, 64, 96
    x = t(rand_array("g9.x", (B, 3, H, W), 1.0)).requires_grad_()
    d = t(rand_array("g9.d", (B, 1, H, W), 0.3, lo=0.05) + np.float32(0.02))
    o.train()
    pred = o(x, d)
    close(pred, g["pred"], 1e-5)
    (pred * t(rand_array("g9.w", pred.shape, 1.0))).sum().backward()
    close(x.grad, g["dx"], 1e-4)
    n = 0
    for k, p in o.named_parameters():
        if (k + "|none") in g:
            assert p.grad is None, k
            continue
        rn = float(g[k + "|norm"][0])
        assert abs(float(p.grad.norm()) - rn) <= 1e-3 * max(rn, 1e-6), k
        n += 1
    assert n > 100
    o.eval()
    with torch.no_grad():
        close(o(x.detach(), d), g["pred_eval"], 1e-5)


def test_g11_metrics():
    g = load("g11_metrics")
    inv = rand_array("g11.inv", (1, 1, 36, 48), 0.3, lo=0.03) + np.float32(0.01)
    gt = rand_array("g11.gt", (60, 80), 60.0, lo=0.0); gt[rand_array("g11.m", gt.shape, 1.0, lo=0.0) < 0.9] = 0
    r = OS.val_metrics(t(inv), gt, (60, 80))
    got = [r[k] for k in ("mae", "rmse", "imae", "irmse", "abs_rel", "sq_rel", "delta1")]
    assert np.allclose(got, g["vals"], rtol=1e-6)
    close(r["pred"], g["pred"], 1e-6)
