"""Generate golden vectors by IMPORTING THE REFERENCE (/root/reference) in the build container.

Run here only:  python tests/golden/make_golden.py
The reference cannot travel to the GPU box; what is committed is data: seeds, shapes, outputs, gradient summaries.
Weights and inputs are regenerated from names+seeds by `tests/golden/fill.py` on both sides, so fixtures stay
small.  torchvision / cv2 / tensorboard are absent here: tests/golden/stubs provides import stubs (torchvision's
roi_pool is the oracle's restatement -- third-party arithmetic that the reference itself cannot pin).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path[:0] = [ROOT, os.path.join(HERE, "stubs"), "/root/reference", "/root/reference/RCNet"]
tb = types.ModuleType("torch.utils.tensorboard")
tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = tb

from tests.golden.fill import fill_state_dict, rand_array  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def grad_summary(named):
    out = {}
    for k, p in named:
        if p.grad is None:
            out[k + "|none"] = np.zeros(1, np.float32)
        else:
            g = p.grad.detach().float().reshape(-1)
            out[k + "|norm"] = np.array([g.norm().item()], np.float32)
            out[k + "|head"] = g[:16].numpy().copy()
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


# ---------------------------------------------------------------------------------------------- G1 / G2
def g_attention():
    import linear_attention as la
    N, L, S, H, D = 4, 21, 21, 8, 16
    q, k, v = [t(rand_array("g1." + n, (N, L if n == "q" else S, H, D), 1.0)).requires_grad_() for n in "qkv"]
    out = la.LinearAttention()(q, k, v)
    w = t(rand_array("g1.w", out.shape, 1.0))
    (out * w).sum().backward()
    save("g1_linear_attention", out=out.detach().numpy(), dq=q.grad.numpy(), dk=k.grad.numpy(), dv=v.grad.numpy())

    layer = la.LoFTREncoderLayer(128, 8)
    fill_state_dict(layer, "g2.layer")
    x = t(rand_array("g2.x", (3, 21, 128), 1.0)).requires_grad_()
    s = t(rand_array("g2.s", (3, 21, 128), 1.0)).requires_grad_()
    o = layer(x, s)
    (o * t(rand_array("g2.w", o.shape, 1.0))).sum().backward()
    save("g2_loftr_layer", out=o.detach().numpy(), dx=x.grad.numpy(), ds=s.grad.numpy(),
         **grad_summary(layer.named_parameters()))

    tf = la.LocalFeatureTransformer(['self', 'cross'], n_layers=4, d_model=128)
    fill_state_dict(tf, "g2.tf")
    a = t(rand_array("g2.a", (2, 21, 128), 1.0)).requires_grad_()
    b = t(rand_array("g2.b", (2, 21, 128), 1.0)).requires_grad_()
    o0, o1 = tf(a, b)
    ((o0 * t(rand_array("g2.w0", o0.shape, 1.0))).sum() + (o1 * t(rand_array("g2.w1", o1.shape, 1.0))).sum()).backward()
    save("g2_transformer", out0=o0.detach().numpy(), out1=o1.detach().numpy(), da=a.grad.numpy(), db=b.grad.numpy(),
         **grad_summary(tf.named_parameters()))


# --------------------------------------------------------------------------------------------------- G3
def g_resnet_encoder():
    import networks
    enc = networks.ResNetEncoder(n_layer=18, input_channels=3, n_filters=[32, 64, 128, 128, 128],
                                 weight_initializer='kaiming_uniform', activation_func='leaky_relu', use_batch_norm=True)
    fill_state_dict(enc, "g3.enc")
    x = t(rand_array("g3.x", (2, 3, 96, 128), 1.0, lo=0.0))
    enc.train()
    latent, skips = enc(x)
    loss = (latent * t(rand_array("g3.wl", latent.shape, 1.0))).sum()
    for i, s in enumerate(skips):
        loss = loss + (s * t(rand_array("g3.ws%d" % i, s.shape, 1.0))).sum() * 0.1
    loss.backward()
    sd = enc.state_dict()
    arrs = dict(latent=latent.detach().numpy(), skip3=skips[3].detach().numpy(),
                skip0_sub=skips[0].detach().numpy()[:, ::4, ::4, ::4].copy(),
                rm=sd['blocks3.0.conv1.batch_norm.running_mean'].numpy(), rv=sd['blocks3.0.conv1.batch_norm.running_var'].numpy(),
                rm1=sd['conv1.batch_norm.running_mean'].numpy(), rv1=sd['conv1.batch_norm.running_var'].numpy())
    arrs.update(grad_summary(enc.named_parameters()))
    enc.eval()
    with torch.no_grad():
        le, _ = enc(x)
    arrs["latent_eval"] = le.numpy()
    save("g3_resnet_encoder", **arrs)


# --------------------------------------------------------------------------------------------------- G5
def g_decoder():
    import networks
    for tag, patch, R in (("small", (64, 32), 2), ("zju", (240, 100), 1)):
        dec = networks.MultiScaleDecoder(input_channels=256, output_channels=1, n_resolution=1,
                                         n_filters=[256, 128, 64, 32, 16], n_skips=[128, 128, 64, 32, 0],
                                         weight_initializer='kaiming_uniform', activation_func='leaky_relu',
                                         output_func='linear', use_batch_norm=True, deconv_type='up')
        fill_state_dict(dec, "g5.dec")
        lh, lw = patch[0] // 32, patch[1] // 32
        sizes = [(int(patch[0] * s), int(patch[1] * s)) for s in (1 / 2., 1 / 4., 1 / 8., 1 / 16.)]
        chans = [32, 64, 128, 128]
        x = t(rand_array("g5.%s.x" % tag, (R, 256, lh, lw), 1.0)).requires_grad_()
        skips = [t(rand_array("g5.%s.s%d" % (tag, i), (R, chans[i]) + sizes[i], 1.0)).requires_grad_() for i in range(4)]
        dec.train()
        out = dec(x, skips, shape=patch)[-1]
        (out * t(rand_array("g5.%s.w" % tag, out.shape, 1.0))).sum().backward()
        arrs = dict(out=out.detach().numpy(), dx=x.grad.numpy(), ds3=skips[3].grad.numpy(),
                    ds0_sub=skips[0].grad.numpy()[:, ::4, ::4, ::4].copy())
        arrs.update(grad_summary(dec.named_parameters()))
        save("g5_decoder_" + tag, **arrs)


# ------------------------------------------------------------------------------------------------- G6 / G10
def g_rcnet_e2e():
    import rcnet_model
    import rcnet_main
    patch = [64, 32]
    model = rcnet_model.RCNetModel(3, 3, patch, ['rcnet', 'batch_norm'], [32, 64, 128, 128, 128], [32, 64, 128, 128, 128],
                                   ['multiscale', 'batch_norm'], [256, 128, 64, 32, 16], device=torch.device('cpu'))
    fill_state_dict(model.encoder, "g6.enc")
    fill_state_dict(model.decoder, "g6.dec")
    B, K, H, W = 2, 3, 64, 96
    pad_y, pad_x = patch[0] // 2, patch[1] // 2
    img = t(rand_array("g6.img", (B, 3, H, W), 1.0, lo=0.0))
    img = torch.nn.functional.pad(img, (pad_x, pad_x, pad_y, pad_y), mode='replicate')
    rs = np.random.RandomState(606)
    pts = np.stack([rs.randint(0, W, (B, K)) + pad_x, rs.randint(0, H, (B, K)) + pad_y, rs.uniform(1.5, 30.0, (B, K))], -1).astype(np.float32)
    boxes = np.stack([pts[..., 0] - pad_x, pts[..., 1] - pad_y, pts[..., 0] + pad_x, pts[..., 1] + pad_y], -1).astype(np.float32)
    gt = rand_array("g6.gt", (B * K, 1, patch[0], patch[1]), 1.0, lo=0.0) * 30.0
    gt[rand_array("g6.gtm", gt.shape, 1.0, lo=0.0) < 0.5] = 0.0
    z = pts[..., 2].reshape(-1)
    for r in range(B * K):  # make some positives
        gt[r, 0, ::3, ::2] = np.where(gt[r, 0, ::3, ::2] > 0, z[r] + 0.2, 0.0)
    gt_t, pts_t = t(gt), t(pts).view(B * K, 3)

    # label build exactly as rcnet_main.train :308-332 (copied call pattern, executed on the reference's ops)
    radar_depth = pts_t[..., 2].view(pts_t.shape[0], 1, 1, 1)
    dist = torch.abs(gt_t - radar_depth * torch.ones_like(gt_t))
    label = torch.where(dist < 0.5, torch.ones_like(gt_t), torch.zeros_like(gt_t))
    label = torch.where(gt_t > 0, label, torch.zeros_like(label))
    valid = torch.where(gt_t <= 0, torch.zeros_like(gt_t), torch.ones_like(gt_t))

    model.train()
    blist = [t(boxes[i]) for i in range(B)]
    logits = model.forward(img, pts_t, blist, return_logits=True)
    loss, _ = model.compute_loss(logits=logits, ground_truth=label.float(), validity_map=valid, w_positive_class=2.5)
    loss.backward()
    arrs = dict(pts=pts, boxes=boxes, logits=logits.detach().numpy(), loss=np.array([loss.item()], np.float32),
                label=label.numpy().astype(np.uint8), valid=valid.numpy().astype(np.uint8))
    arrs.update({"enc." + k: v for k, v in grad_summary(model.encoder.named_parameters()).items()})
    arrs.update({"dec." + k: v for k, v in grad_summary(model.decoder.named_parameters()).items()})
    save("g6_rcnet_e2e", **arrs)

    # G10: the reference's own forward_output (rcnet_main.py:435-487) in eval mode on one unpadded image
    model.eval()
    with torch.no_grad():
        img1 = t(rand_array("g10.img", (1, 3, H, W), 1.0, lo=0.0))
        N = 7
        rs = np.random.RandomState(1010)
        p1 = np.stack([rs.randint(0, W, N) + pad_x, rs.randint(0, H, N) + pad_y, rs.uniform(1.5, 30.0, N)], -1).astype(np.float32)
        b1 = np.stack([p1[:, 0] - pad_x, p1[:, 1] - pad_y, p1[:, 0] + pad_x, p1[:, 1] + pad_y], -1).astype(np.float32)
        depth, resp = rcnet_main.forward_output(model, img1, t(p1), [t(b1)], response_thr=0.5, device=torch.device('cpu'))
        img1p = torch.nn.functional.pad(img1, (pad_x, pad_x, pad_y, pad_y), mode='replicate')
        crops = model.forward(img1p, t(p1), [t(b1)], return_logits=False)
        thr = float(np.median(crops.numpy()))  # a threshold that splits the responses
        depth2, resp2 = rcnet_main.forward_output(model, img1, t(p1), [t(b1)], response_thr=thr, device=torch.device('cpu'))
    save("g10_forward_output", pts=p1, boxes=b1, crops=crops.numpy(), depth=depth.numpy(), resp=resp.numpy(),
         thr2=np.array([thr], np.float32), depth2=depth2.numpy(), resp2=resp2.numpy())


# ---------------------------------------------------------------------------------------------- G12: depth PNG codec
def g_depth_png():
    """The reference's own save_depth / load_depth (data/data_utils.py:94-143) on a seeded depth map with the edge values of the uint16 * 256
    encoding: the PNG file it writes (bytes), what it reads back, and the stored integers."""
    import io
    import tempfile
    from data import data_utils
    z = rand_array("g12.z", (37, 53), 90.0, lo=0.0).astype(np.float32)
    z[rand_array("g12.m", z.shape, 1.0, lo=0.0) < 0.6] = 0.0
    z[0, :8] = np.array([0.0, 0.001, 0.0039, 0.00390625, 1.5, 255.99, 255.998, 99.999], np.float32)
    z[1, :4] = np.array([256.0, 300.0, 1e-9, 255.99609375], np.float32)      # >= 256 m saturates at 65535 in the reference's file
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "z.png")
        data_utils.save_depth(z, path)
        raw = np.frombuffer(open(path, "rb").read(), dtype=np.uint8).copy()
        back = data_utils.load_depth(path)
        from PIL import Image
        stored = np.array(Image.open(io.BytesIO(raw.tobytes()))).astype(np.uint16)
    save("g12_depth_png", z=z, png=raw, loaded=back.astype(np.float32), stored=stored)


# ---------------------------------------------------------------------------------------------- G13: batch transforms
def g_transforms():
    """The reference's own Transforms.transform (RCNet/rcnet_transforms.py:58-240) with the ZJU training configuration
    (train_rcnet_zju.py:52-59) on a seeded batch, torch seeded so that the draws can be repeated; torchvision's photometric functions are
    the oracle's restatement (stub).  Stores inputs that are not regenerated by name, the outputs, and the decisions the class made."""
    import rcnet_transforms
    B, K, H, W, ph, pw = 4, 3, 40, 52, 12, 8
    image = np.floor(rand_array("g13.img", (B, 3, H, W), 256.0, lo=0.0)).astype(np.float32)
    labels = rand_array("g13.lab", (B, K, 1, ph, pw), 30.0, lo=0.0)
    xs = np.floor(rand_array("g13.bx", (B, K), W - pw, lo=0.0)) + pw // 2
    ys = np.floor(rand_array("g13.by", (B, K), H - ph, lo=0.0)) + ph // 2
    boxes = np.stack([xs - pw // 2, ys - ph // 2, xs + pw // 2, ys + ph // 2], -1).astype(np.float32)
    points = np.stack([xs, ys, rand_array("g13.z", (B, K), 50.0, lo=0.1)], -1).astype(np.float32)
    tr = rcnet_transforms.Transforms(normalized_image_range=[0, 1], random_brightness=[0.80, 1.20], random_contrast=[0.80, 1.20],
                                     random_saturation=[0.80, 1.20], random_noise_type='none', random_noise_spread=-1,
                                     random_flip_type=['horizontal'])
    torch.manual_seed(1313)
    [img_o], [lab_o], [pts_o], [box_o] = tr.transform(images_arr=[t(image.copy())], labels_arr=[t(labels.copy())], points_arr=[t(points.copy())],
                                                      bounding_boxes_arr=[t(boxes.copy())], random_transform_probability=1.00)
    save("g13_transforms", image=image.astype(np.uint8), boxes=boxes, points=points, out_image=img_o.numpy(), out_labels=lab_o.numpy(),
         out_points=pts_o.numpy(), out_boxes=box_o.numpy(), seed=np.array([1313]))


def g_transforms_all():
    """G13b: the same class with EVERY augmentation it has switched on -- gaussian point noise (:157-166, add_noise :398-432) and both flips
    (:168-217, including the vertical flip's box update that indexes boxes 1 and 3 of the sample) -- K = 5 boxes per sample so that the
    reference's own indexing is in range."""
    import rcnet_transforms
    B, K, H, W, ph, pw = 6, 5, 40, 52, 12, 8
    image = np.floor(rand_array("g13b.img", (B, 3, H, W), 256.0, lo=0.0)).astype(np.float32)
    labels = rand_array("g13b.lab", (B, K, 1, ph, pw), 30.0, lo=0.0)
    xs = np.floor(rand_array("g13b.bx", (B, K), W - pw, lo=0.0)) + pw // 2
    ys = np.floor(rand_array("g13b.by", (B, K), H - ph, lo=0.0)) + ph // 2
    boxes = np.stack([xs - pw // 2, ys - ph // 2, xs + pw // 2, ys + ph // 2], -1).astype(np.float32)
    points = np.stack([xs, ys, rand_array("g13b.z", (B, K), 50.0, lo=0.1)], -1).astype(np.float32)
    out = {}
    for tag, kind, spread, seed in (("gauss", "gaussian", 0.75, 1314), ("unif", "uniform", 2.0, 1316)):
        tr = rcnet_transforms.Transforms(normalized_image_range=[0, 1], random_brightness=[0.80, 1.20], random_contrast=[0.80, 1.20],
                                         random_saturation=[0.80, 1.20], random_noise_type=kind, random_noise_spread=spread,
                                         random_flip_type=['horizontal', 'vertical'])
        torch.manual_seed(seed)
        [img_o], [lab_o], [pts_o], [box_o] = tr.transform(images_arr=[t(image.copy())], labels_arr=[t(labels.copy())], points_arr=[t(points.copy())],
                                                          bounding_boxes_arr=[t(boxes.copy())], random_transform_probability=1.00)
        out.update({tag + "_out_image": img_o.numpy(), tag + "_out_labels": lab_o.numpy(), tag + "_out_points": pts_o.numpy(),
                    tag + "_out_boxes": box_o.numpy(), tag + "_seed": np.array([seed]), tag + "_spread": np.array([spread], np.float32)})
    save("g13b_transforms_all", image=image.astype(np.uint8), boxes=boxes, points=points, **out)


# ---------------------------------------------------------------------------------------------- G14: projection + scatter
def g_projection():
    """The reference's own project_pcl_to_image + min_max_filter (data/preprocess/project_transform.py) on a seeded point cloud with
    KITTI-like float64 calibration matrices, followed by the scatter loop of pointcloud_project_zju.py:57-66 (depth_map[v, u] = max(depth, 1)
    in the returned far-to-near order).  Many points share pixels, some project outside, behind the camera or out of range."""
    sys.path.insert(0, "/root/reference/data/preprocess")
    import project_transform as PT
    rs = np.random.RandomState(1414)
    n, H, W = 6000, 48, 64
    pts = np.concatenate([rs.uniform(-12, 12, (n, 1)), rs.uniform(-4, 4, (n, 1)), rs.uniform(-5, 120, (n, 1)), rs.uniform(0, 1, (n, 2))], 1).astype(np.float32)
    ang = 0.03
    T = np.array([[np.cos(ang), 0, np.sin(ang), 0.1], [0, 1, 0, -0.2], [-np.sin(ang), 0, np.cos(ang), 0.3], [0, 0, 0, 1]], np.float64)
    P = np.array([[60.0, 0, 32.0, 0], [0, 60.0, 24.0, 0], [0, 0, 1.0, 0], [0, 0, 0, 1.0]], np.float64)
    uvs, depth = PT.project_pcl_to_image(point_cloud=pts, t_camera_pcl=T, camera_projection_matrix=P, image_shape=(H, W, 3))
    idx = PT.min_max_filter(depth, max_value=100.0, min_value=1.5)
    uvs, depth = uvs[idx], depth[idx]
    depth_map = np.zeros((H, W), dtype=np.float32)
    for i in range(len(depth)):            # pointcloud_project_zju.py:61-64
        u, v = uvs[i]
        depth_map[v, u] = max(depth[i], 1)
    save("g14_projection", points=pts, T=T, P=P, depth_map=depth_map, uvs=uvs.astype(np.int32), depth=depth.astype(np.float64))


# ---------------------------------------------------------------------------------------------- G15: lidar interpolation
def g_interpolation():
    """The reference's own interpolate_depth (data/data_utils.py:231-275; linear and log space) and interpolate_depth_delft (:333-367,
    validity_map=None) on a seeded sparse depth map: scattered valid pixels plus a short collinear run and a 2x2 cluster (co-circular
    points, the case where the Delaunay triangulation is not unique)."""
    from data import data_utils
    H, W = 40, 56
    z = np.zeros((H, W), np.float32)
    rs = np.random.RandomState(1515)
    idx = rs.choice(H * W, 90, replace=False)
    z.flat[idx] = rs.uniform(1.5, 80.0, idx.size).astype(np.float32)
    z[20, 10:15] = np.array([5.0, 6.0, 7.5, 9.0, 12.0], np.float32)     # collinear
    z[30:32, 40:42] = np.array([[20.0, 21.0], [22.0, 23.5]], np.float32)  # co-circular
    valid = (z > 0).astype(np.float32)
    lin = data_utils.interpolate_depth(z, valid, log_space=False)
    logz = data_utils.interpolate_depth(z, valid, log_space=True)
    delft = data_utils.interpolate_depth_delft(z)
    save("g15_interpolation", depth=z, linear=lin.astype(np.float64), log=logz.astype(np.float64), delft=delft.astype(np.float64))


# ---------------------------------------------------------------------------------------------- G16: training samples (fake radar)
def g_datasets():
    """The reference's own RCNetTrainingDataset.__getitem__ (data/datasets.py:168-291) on files written here: an RGB frame (PIL), a radar
    .npy and a 16-bit ground-truth PNG (the reference's save_depth).  Three seeded draws: plain radar sampling, the fake-radar branch
    (sample_probability_of_lidar = 1: points drawn from the lidar map + noise, :214-240), and a frame with fewer points than are sampled
    (repeat x100 branch) whose radar file is a single (3,) row."""
    import random
    import tempfile
    from PIL import Image
    from data import datasets as RD
    from data import data_utils as RU
    H, W, ph, pw, K = 36, 60, 12, 8, 6
    rs = np.random.RandomState(1616)
    img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    gt = np.zeros((H, W), np.float32)
    idx = rs.choice(H * W, 500, replace=False)
    gt.flat[idx] = rs.uniform(0.5, 70.0, idx.size).astype(np.float32)
    radar = np.stack([rs.uniform(0, W - 1, 11), rs.uniform(0, H - 1, 11), rs.uniform(2, 60, 11)], 1)          # float64 (x, y, depth)
    single = radar[3].copy()
    out = dict(image=img, radar=radar, single=single)
    with tempfile.TemporaryDirectory() as d:
        ip, rp, sp, gp = [os.path.join(d, n) for n in ("img.png", "radar.npy", "single.npy", "gt.png")]
        Image.fromarray(img).save(ip)
        np.save(rp, radar); np.save(sp, single)
        RU.save_depth(gt, gp)
        out["gt_png"] = np.frombuffer(open(gp, "rb").read(), dtype=np.uint8)
        out["img_png"] = np.frombuffer(open(ip, "rb").read(), dtype=np.uint8)
        for tag, rpath, p, seed in (("plain", rp, 0.0, 5), ("fake", rp, 1.0, 6), ("few", sp, 0.0, 7), ("fake_few", sp, 1.0, 8)):
            ds = RD.RCNetTrainingDataset([ip], [rpath], [gp], patch_size=[ph, pw], total_points_sampled=K, sample_probability_of_lidar=p)
            np.random.seed(seed); random.seed(seed)
            im, pts, boxes, crops = ds[0]
            out.update({tag + "_image": im, tag + "_points": pts, tag + "_boxes": boxes, tag + "_crops": crops, tag + "_seed": np.array([seed])})
    out["cfg"] = np.array([ph, pw, K])
    save("g16_datasets", **out)


# ---------------------------------------------------------------------------------------------- G17: knot interpolation (Interpolator2D)
def g_interpolator():
    """The reference's own Interpolator2D (modules/interpolator.py:21-49) on a seeded inverse-depth pair: interpolated scale maps for
    'linear' and 'nearest' (scipy griddata, fill_value 1.0).  `tie` marks the pixels with two equidistant nearest knots, the one thing
    scipy's cKDTree leaves implementation-defined."""
    from modules import interpolator as RI
    H, W = 34, 46
    rs = np.random.RandomState(1717)
    pred_inv = rs.uniform(0.02, 0.4, (H, W)).astype(np.float32)
    sparse_inv = np.zeros((H, W), np.float32)
    valid = np.zeros((H, W), bool)
    valid.flat[rs.choice(H * W, 40, replace=False)] = True
    r, c = np.nonzero(valid)
    qr, qc = np.mgrid[0:H, 0:W]
    srt = np.sort((qr[..., None] - r) ** 2 + (qc[..., None] - c) ** 2, axis=-1)
    tie = srt[..., 0] == srt[..., 1]          # pixels with two equidistant nearest knots: scipy's cKDTree leaves the pick implementation-defined
    sparse_inv[valid] = rs.uniform(0.02, 0.4, int(valid.sum())).astype(np.float32)
    it = RI.Interpolator2D(pred_inv=pred_inv, sparse_depth_inv=sparse_inv, valid=valid)
    it.generate_interpolated_scale_map(interpolate_method='linear', fill_corners=False)
    lin = it.interpolated_scale_map.copy()
    it.generate_interpolated_scale_map(interpolate_method='nearest', fill_corners=False)
    near = it.interpolated_scale_map.copy()
    save("g17_interpolator", pred_inv=pred_inv, sparse_inv=sparse_inv, valid=valid, linear=lin, nearest=near, tie=tie, knot_coords=it.knot_coords,
         knot_scales=it.knot_scales, knot_shifts=it.knot_shifts)


if __name__ == "__main__":
    which = sys.argv[1:] or ["attention", "resnet", "decoder", "e2e", "png", "transforms", "projection", "interpolation", "datasets", "interpolator"]
    if "datasets" in which:
        g_datasets()
    if "interpolator" in which:
        g_interpolator()
    if "interpolation" in which:
        g_interpolation()
    if "projection" in which:
        g_projection()
    if "transforms" in which:
        g_transforms()
    if "transforms_all" in which or "transforms" in which:
        g_transforms_all()
    if "png" in which:
        g_depth_png()
    if "attention" in which:
        g_attention()
    if "resnet" in which:
        g_resnet_encoder()
    if "decoder" in which:
        g_decoder()
    if "e2e" in which:
        g_rcnet_e2e()
