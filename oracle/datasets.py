"""ORACLE (test infrastructure only): the reference's RC-Net training sample construction restated point by point --
data/datasets.py RCNetTrainingDataset.__getitem__ :168-291 (radar sampling :203-206, fake radar from lidar :214-240, pad shift + boxes
:243-255, ground-truth crops :257-275).  Random draws in the reference's order from the global numpy / python generators.
Pinned by tests/golden/g16_datasets.npz, produced by the reference class itself (tests/golden/make_golden.py g_datasets)."""
import random

import numpy as np


def training_sample(image_chw, radar_points, ground_truth_chw, patch_size, total_points_sampled, sample_probability_of_lidar):
    pad_x, pad_y = patch_size[1] // 2, patch_size[0] // 2
    padding = ((0, 0), (pad_y, pad_y), (pad_x, pad_x))
    image = np.pad(image_chw, pad_width=padding, mode='edge')                       # :176-179
    pts = np.asarray(radar_points)
    if pts.ndim == 1:                                                               # :195-197
        pts = np.expand_dims(pts, axis=0)
    if pts.shape[0] <= total_points_sampled:                                        # :203-204
        pts = np.repeat(pts, 100, axis=0)
    pts = pts[np.random.randint(pts.shape[0], size=total_points_sampled), :]        # :205-206
    gt = ground_truth_chw
    if random.random() < sample_probability_of_lidar:                               # :214
        g2 = np.copy(gt).squeeze()
        where = np.where(g2 > 1)                                                    # :218
        ri = random.sample(range(0, len(where[0])), total_points_sampled)           # :221
        px, py = where[1][ri], where[0][ri]
        pz = g2[py, px]
        nx = np.random.normal(0, 25, pts.shape[0])                                  # :227
        nz = np.random.uniform(low=0.0, high=0.5, size=pts.shape[0])                # :228
        fake = np.copy(pts)
        for i in range(fake.shape[0]):
            fake[i, 0] = min(max(px[i] + nx[i], 0), g2.shape[1])                    # :231-232
            fake[i, 2] = pz[i] + nz[i]                                              # :233
            fake[i, 0] = int(fake[i, 0])                                            # :238
            fake[i, 1] = int(fake[i, 1])                                            # :239
        pts = np.copy(fake)
    else:
        pts = np.copy(pts)
    boxes = []
    for i in range(pts.shape[0]):                                                   # :243-255
        pts[i, 0] = pts[i, 0] + pad_x
        pts[i, 1] = pts[i, 1] + pad_y
        boxes.append(np.asarray([pts[i, 0] - pad_x, pts[i, 1] - pad_y, pts[i, 0] + pad_x, pts[i, 1] + pad_y]))
    gt = np.pad(gt, pad_width=padding, mode='constant', constant_values=0)          # :257-261
    crops = []
    for i in range(pts.shape[0]):                                                   # :266-275
        sx, ex = int(pts[i, 0] - pad_x), int(pts[i, 0] + pad_x)
        sy, ey = int(pts[i, 1] - pad_y), int(pts[i, 1] + pad_y)
        crops.append(gt[:, sy:ey, sx:ex])
    return (image.astype(np.float32), pts.astype(np.float32), np.stack([b.astype(np.float32) for b in boxes], axis=0),
            np.asarray(crops).astype(np.float32))
