"""Scale-Map-Learner training / validation harness on MI355X: counterpart of the reference's train_zju.py step body
(:246-392) and val_zju.py (:124-254) with the per-sample CPU pre-step (D2H + scipy + numpy + H2D per sample) replaced by
batched device kernels.  Also the synthetic batch generator restating the tensor contract of data/UTV_dataset.py:157-224.
"""
import ctypes

import numpy as np
import torch

from . import engine
from .loss import compute_loss
from .midas.midas_net_custom import MidasNet_small_videpth
from .net_utils import OutlierRemoval

# train_zju.py:429-487
ZJU_SML_CONFIG = dict(min_pred=0.1, max_pred=255.0, min_depth=0.0, max_depth=100.0, learning_rate=1e-4, loss_func='l1', w_smoothness=0.2,
                      w_lidar_loss=1.5, w_edge=0.0, sobel_filter_size=7, outlier_removal_kernel_size=3, outlier_removal_threshold=1.5,
                      scale_bounds=(0.01, 0.3), mean_std=dict(int_depth=(0.729, 0.210), int_scales=(0.404, 0.117)), interp='rcnet', global_alignment='s')


def net_size(height, width, net_h=288, net_w=384, multiple=32):
    """modules/midas/transforms.py:62-131 Resize.get_size with keep_aspect_ratio, 'minimal', ensure_multiple_of=32."""
    sh, sw = net_h / height, net_w / width
    if abs(1 - sw) < abs(1 - sh):
        sh = sw
    else:
        sw = sh
    r = lambda x: int(np.round(x / multiple) * multiple)  # noqa: E731
    return r(sh * height), r(sw * width)


def build_model(device, cfg=ZJU_SML_CONFIG):
    return MidasNet_small_videpth(device=device, in_channels=3, min_pred=cfg['min_pred'], max_pred=cfg['max_pred'])


def synthetic_batch(batch_size, height=288, width=384, seed=1234, device='cpu'):
    """(image (B,3,H,W) in [0,1], mono inverse depth (B,1,H,W), sparse radar depth, dense gt, sparse lidar gt, rcnet quasi-dense depth)."""
    g = torch.Generator().manual_seed(seed)
    B, H, W = batch_size, height, width
    image = torch.rand((B, 3, H, W), generator=g)
    depth = torch.rand((B, 1, H, W), generator=g) * 60.0 + 2.0
    s_true = 0.03 + 0.1 * torch.rand((B, 1, 1, 1), generator=g)
    mono = 1.0 / (depth * s_true) * (1.0 + 0.05 * torch.randn((B, 1, H, W), generator=g))
    radar = depth * (torch.rand((B, 1, H, W), generator=g) < 0.002).float()
    rcnet = depth * (torch.rand((B, 1, H, W), generator=g) < 0.05).float() * (1.0 + 0.02 * torch.randn((B, 1, H, W), generator=g))
    gt = depth * (torch.rand((B, 1, H, W), generator=g) < 0.7).float()
    sparse_gt = depth * (torch.rand((B, 1, H, W), generator=g) < 0.02).float()
    return tuple(t.to(device).contiguous() for t in (image, mono, radar, gt, sparse_gt, rcnet))


def prepare_inputs(image, mono_pred, sparse_depth, rcnet, net_hw, cfg=ZJU_SML_CONFIG):
    """S1 on device: valid masks, inverse depth, global alignment ('s': bounded L1 scale fit, estimator.py:146-160; 'st': closed-form
    least-squares scale + shift, estimator.py:5-29), int_depth / int_scales, min-max normalise, nearest resize, mean/std normalise, gray
    image.  Returns x (B,3,h,w) logical NCHW (channels_last, fp32), d (B,1,h,w), scale (B,) -- or (scale, shift) for 'st'."""
    lib = engine.L()
    B, _, H, W = image.shape
    h, w = net_hw
    st = engine._stream(image)
    p = engine._p
    scale = torch.empty(B, dtype=torch.float32, device=image.device)
    nvalid = torch.empty(B, dtype=torch.int32, device=image.device)
    shift = None
    mode = cfg.get('global_alignment', 's')
    if mode == 'st':
        shift = torch.empty(B, dtype=torch.float32, device=image.device)
        engine._chk(lib.rd_sml_scale_shift_ls(p(mono_pred), p(sparse_depth), B, H * W, cfg['min_depth'], cfg['max_depth'], p(scale), p(shift),
                                              p(nvalid), st), "rd_sml_scale_shift_ls")
    elif mode == 's':
        lo, hi = cfg['scale_bounds']
        engine._chk(lib.rd_sml_scale_align(p(mono_pred), p(sparse_depth), B, H * W, cfg['min_depth'], cfg['max_depth'], lo, hi, p(scale), p(nvalid),
                                           st), "rd_sml_scale_align")
    else:
        raise NotImplementedError("global_alignment %r (train_zju.py:278-302 knows 's' and 'st')" % (mode,))
    mm = torch.empty((B, 3), dtype=torch.float32, device=image.device)
    x = torch.empty((B, h, w, 3), dtype=torch.float32, device=image.device)
    d = torch.empty((B, 1, h, w), dtype=torch.float32, device=image.device)
    use_rc = 1 if ('rcnet' in cfg['interp'] and rcnet is not None) else 0
    (m0, s0), (m1, s1) = cfg['mean_std']['int_depth'], cfg['mean_std']['int_scales']
    engine._chk(lib.rd_sml_build_inputs(p(image), p(mono_pred), p(sparse_depth), p(rcnet), p(scale), p(shift), p(mm), B, H, W, h, w,
                                        cfg['min_depth'], cfg['max_depth'], 1.0 / cfg['min_pred'], 1.0 / cfg['max_pred'], use_rc, m0, s0, m1, s1,
                                        p(x), p(d), st), "rd_sml_build_inputs")
    return x.permute(0, 3, 1, 2), d, (scale if shift is None else (scale, shift))


def nearest_resize(t, h, w):
    """cv2 INTER_NEAREST resize of (B,1,H,W) maps (gt / sparse gt) through the nearest-upsample kernel's index rule."""
    B, C, H, W = t.shape
    if (H, W) == (h, w):
        return t
    ys = torch.clamp((torch.arange(h, device=t.device, dtype=torch.float64) * (H / h)).floor().long(), max=H - 1)
    xs = torch.clamp((torch.arange(w, device=t.device, dtype=torch.float64) * (W / w)).floor().long(), max=W - 1)
    return t[:, :, ys][:, :, :, xs].contiguous()  # data preparation (index gather), off the timed path when sizes already match


def forward_loss(model, batch, cfg=ZJU_SML_CONFIG, outlier=None):
    """Pre-step + forward + loss of one SML step (train_zju.py:246-376) -> loss tensor."""
    image, mono, sparse_depth, gt, sparse_gt, rcnet = batch
    H, W = image.shape[-2:]
    # network input size: the reference's transform rule (nearest resize to the multiple of 32 closest to 288, modules/midas/transforms.py:62-131
    # through train_zju.py:164) unless cfg['net_hw'] asks for a size outright (BASELINE configs[4]: the 512x1024 frames at native resolution)
    hw = tuple(cfg['net_hw']) if cfg.get('net_hw') else net_size(H, W)
    x, d, _ = prepare_inputs(image, mono, sparse_depth, rcnet, hw, cfg)
    gt_r, sgt_r = nearest_resize(gt, *hw), nearest_resize(sparse_gt, *hw)
    pred = model.forward(x, d)
    lib = engine.L()
    d_depth = torch.empty_like(d)
    engine._chk(lib.rd_reciprocal(engine._p(d), None, engine._p(d_depth), d.numel(), engine._stream(d)), "rd_reciprocal")
    sml_depth = engine.run_region(lambda p: engine.reciprocal(p), (pred,), [])
    if outlier is not None:
        gt_r = outlier.remove_outliers(gt_r)
    loss, info = compute_loss(image=d_depth, output_depth=sml_depth, gt_interp=gt_r, gt_sparse=sgt_r, loss_func=cfg['loss_func'],
                              w_smoothness=cfg['w_smoothness'], sobel_filter_size=cfg['sobel_filter_size'],
                              validity_map_loss_smoothness=None, w_lidar_loss=cfg['w_lidar_loss'], w_edge=cfg['w_edge'],
                              invalid_map_gt=None, w_unsupervised=0.0)
    return loss


def compute_gradients(model, optimizer, batch, cfg=ZJU_SML_CONFIG, outlier=None, loss_scale=1.0):
    from .rcnet_main import scaled_backward
    loss = forward_loss(model, batch, cfg, outlier)
    optimizer.zero_grad()
    scaled_backward(loss, optimizer, loss_scale)
    return loss


def train_step(model, optimizer, batch, cfg=ZJU_SML_CONFIG, reducer=None, outlier=None, loss_scale=1.0):
    loss = compute_gradients(model, optimizer, batch, cfg, outlier, loss_scale)
    if reducer is not None:
        reducer.reduce()
    optimizer.step()
    return loss


def make_outlier_removal(cfg=ZJU_SML_CONFIG):
    return OutlierRemoval(cfg['outlier_removal_kernel_size'], cfg['outlier_removal_threshold'])


def validate_batch(model, batch, cfg=ZJU_SML_CONFIG, min_depth_val=0.0, max_depth_val=50.0):
    """val_zju.py:124-254 for a batch: pre-step, forward (no grad), 1/pred -> bicubic to the input size, masked metric sums.
    Returns a dict of per-image metrics (numpy) incl. 'abs_rel'."""
    image, mono, sparse_depth, gt, sparse_gt, rcnet = batch
    B, _, H, W = image.shape
    hw = net_size(H, W)
    lib, p = engine.L(), engine._p
    with torch.no_grad():
        x, d, _ = prepare_inputs(image, mono, sparse_depth, rcnet, hw, cfg)
        pred = model.forward(x, d)
        st = engine._stream(pred)
        depth = torch.empty_like(pred)
        engine._chk(lib.rd_reciprocal(p(pred), None, p(depth), pred.numel(), st), "rd_reciprocal")
        up = torch.empty((B, 1, H, W), dtype=torch.float32, device=pred.device)
        engine._chk(lib.rd_bicubic_resize(p(depth), p(up), B, hw[0], hw[1], H, W, st), "rd_bicubic_resize")
        res = torch.empty((B, 8), dtype=torch.float64, device=pred.device)
        sg = sparse_gt if sparse_gt.is_contiguous() else sparse_gt.contiguous()
        engine._chk(lib.rd_depth_metrics(p(up), p(sg), B, H * W, ctypes.c_float(min_depth_val), ctypes.c_float(max_depth_val), p(res), st),
                    "rd_depth_metrics")
    r = res.cpu().numpy()
    n = np.maximum(r[:, 0], 1.0)
    return dict(count=r[:, 0], mae=r[:, 1] / n, rmse=np.sqrt(r[:, 2] / n), imae=r[:, 3] / n, irmse=np.sqrt(r[:, 4] / n), abs_rel=r[:, 5] / n,
                sq_rel=r[:, 6] / n, delta1=r[:, 7] / n, depth=up)


class GraphedTrainStep(object):
    """SML training step with pre-step + forward + loss + backward replayed from hipGraphs split at the stage marks."""

    def __new__(cls, model, optimizer, batch, cfg=ZJU_SML_CONFIG, reducer=None, outlier=None, warmup=2, loss_scale=1.0):
        from .rcnet_main import GraphedStep
        from .midas.efficientnet_lite3 import _Counted
        counted = [m for m in model.modules() if isinstance(m, _Counted)]

        def bump(delta):
            if model.training:
                for m in counted:
                    m._pending += delta
                model._first_pending += delta
                model.pretrained.layer1._stem_pending = getattr(model.pretrained.layer1, "_stem_pending", 0) + delta
        buffers = [b for b in model.buffers() if b.is_floating_point()]
        step = GraphedStep(lambda: forward_loss(model, batch, cfg, outlier), optimizer, reducer, warmup, bump, buffers, loss_scale)
        step.static_batch = tuple(batch)      # GraphedStep.load_batch(new_batch) copies a new batch into these before a replay
        return step
