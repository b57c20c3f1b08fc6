"""RC-Net batch transforms on MI355X; same class, constructor and transform() signature as the reference's RCNet/rcnet_transforms.py
(Transforms :5, transform :58).  The random decisions are drawn on the host in the reference's order -- do_random_transform, then per
enabled augmentation a coin and a value, the point-noise coin followed by one randn / rand draw per flagged sample and tensor, the
horizontal and the vertical flip coin (:99-217) -- so a run seeded like a CPU run of the reference makes the same decisions; the image
arithmetic, the flips of image / ground-truth crops / boxes and the normalisation are three HIP launches (rd_augment_gray_partials,
rd_augment_image, rd_augment_flip_labels; + rd_augment_vflip_boxes, + one rd_add per noisy point tensor) instead of per-sample Python loops.
train_rcnet_zju.py:57-59 itself runs without point noise and with the horizontal flip only.
"""
import torch

from . import engine


class Transforms(object):
    def __init__(self, normalized_image_range=[0, 255], random_brightness=[-1], random_contrast=[-1], random_saturation=[-1],
                 random_noise_type='none', random_noise_spread=-1, random_flip_type=['none']):
        self.normalized_image_range = normalized_image_range
        self.do_random_brightness = True if -1 not in random_brightness else False
        self.random_brightness = random_brightness
        self.do_random_contrast = True if -1 not in random_contrast else False
        self.random_contrast = random_contrast
        self.do_random_saturation = True if -1 not in random_saturation else False
        self.random_saturation = random_saturation
        self.do_random_noise = True if (random_noise_type != 'none' and random_noise_spread > -1) else False
        self.random_noise_type = random_noise_type
        self.random_noise_spread = random_noise_spread
        self.do_random_horizontal_flip = True if 'horizontal' in random_flip_type else False
        self.do_random_vertical_flip = True if 'vertical' in random_flip_type else False
        self.noise = []

    def draw(self, n_batch, random_transform_probability, points_shapes=()):
        """(n_batch, 8) float32 parameter rows, random numbers drawn exactly as transform() :99-217 draws them.  With point noise enabled,
        `points_shapes` (the shapes of the tensors of points_arr) is needed: add_noise :398-432 draws one randn / rand of the sample's shape
        per flagged sample, tensor by tensor, between the saturation draws and the flip coins; the scaled noise (zeros for the samples
        left alone) is left in self.noise, one host tensor per entry of points_arr."""
        p = torch.zeros((n_batch, 8), dtype=torch.float32)
        do = torch.rand(n_batch) <= random_transform_probability
        for col, enabled, rng in ((0, self.do_random_brightness, self.random_brightness), (2, self.do_random_contrast, self.random_contrast),
                                  (4, self.do_random_saturation, self.random_saturation)):
            if enabled:
                flag = torch.logical_and(do, torch.rand(n_batch) <= 0.50)
                values = torch.rand(n_batch)
                lo, hi = rng
                p[:, col] = flag.float()
                p[:, col + 1] = (hi - lo) * values + lo
        self.noise = []
        if self.do_random_noise:
            flag = torch.logical_and(do, torch.rand(n_batch) <= 0.50)
            for shape in points_shapes:
                noise = torch.zeros(tuple(shape), dtype=torch.float32)
                for b in range(n_batch):
                    if flag[b]:
                        if self.random_noise_type == 'gaussian':
                            noise[b] = self.random_noise_spread * torch.randn(*shape[1:])
                        elif self.random_noise_type == 'uniform':
                            noise[b] = self.random_noise_spread * (torch.rand(*shape[1:]) - 0.5)
                        else:
                            raise ValueError('Unsupported noise type: {}'.format(self.random_noise_type))
                self.noise.append(noise if bool(flag.any()) else None)
        if self.do_random_horizontal_flip:
            p[:, 6] = torch.logical_and(do, torch.rand(n_batch) <= 0.50).float()
        if self.do_random_vertical_flip:
            p[:, 7] = torch.logical_and(do, torch.rand(n_batch) <= 0.50).float()
        return p

    def transform(self, images_arr, labels_arr=[], points_arr=[], bounding_boxes_arr=[], random_transform_probability=0.00, params=None, noise=None):
        """`params` / `noise`: decisions drawn earlier by draw() (tests; a loader thread drawing ahead of the step); default: drawn here."""
        if len(images_arr) != 1 or len(labels_arr) > 1 or len(bounding_boxes_arr) > 1:
            raise NotImplementedError("the RIDERS loops pass one image / label / box tensor (rcnet_main.py:285-290, run_rcnet_zju.py:236-240)")
        image = images_arr[0]
        if image.dim() != 4 or image.shape[1] != 3:
            raise ValueError('Unsupported image shape: {}'.format(tuple(image.shape)))
        B, _, H, W = image.shape
        dev = image.device
        lib, p_, st = engine.L(), engine._p, engine._stream(image)
        if params is None:
            params = self.draw(B, random_transform_probability, [tuple(pt.shape) for pt in points_arr])
            noise = self.noise
        elif noise is None:
            noise = [None] * len(points_arr)
        if self.do_random_noise and len(noise) != len(points_arr):
            raise ValueError("draw() was given %d point shapes, transform() %d point tensors" % (len(noise), len(points_arr)))
        pd = params.to(dev, non_blocking=True)
        img = image if (image.dtype == torch.float32 and image.is_contiguous()) else image.float().contiguous()
        rng = list(self.normalized_image_range)
        if rng == [0, 1]:
            scale, shift = 1.0 / 255.0, 0.0
        elif rng == [0, 255]:
            scale, shift = 1.0, 0.0
        elif rng == [-1, 1]:
            raise NotImplementedError("normalized_image_range [-1, 1] is not used by the RIDERS scripts")
        else:
            raise ValueError('Unsupported normalization range: {}'.format(self.normalized_image_range))
        partial = torch.empty((B, 32), dtype=torch.int64, device=dev)
        engine._chk(lib.rd_augment_gray_partials(p_(img), B, H, W, p_(pd), p_(partial), st), "rd_augment_gray_partials")
        out = torch.empty((B, H, W, 3), dtype=engine.act_dtype(), device=dev)
        engine._chk(lib.rd_augment_image(p_(img), B, H, W, p_(pd), p_(partial), p_(out), engine.rd_of(out), scale, shift, st), "rd_augment_image")
        outputs = [[out.permute(0, 3, 1, 2)]]      # logical NCHW, channels_last: what the network's first layer consumes without a copy
        boxes = None
        flips = self.do_random_horizontal_flip or self.do_random_vertical_flip
        if len(bounding_boxes_arr) > 0:
            boxes = bounding_boxes_arr[0]
            if not (boxes.dtype == torch.float32 and boxes.is_contiguous()):
                boxes = boxes.float().contiguous()
            if self.do_random_vertical_flip and boxes.shape[1] < 4:
                raise IndexError("the vertical flip indexes boxes 1 and 3 of each sample (rcnet_transforms.py:213-217): %d boxes" % boxes.shape[1])
        if len(labels_arr) > 0:
            lab = labels_arr[0]
            lab = lab if (lab.dtype == torch.float32 and lab.is_contiguous()) else lab.float().contiguous()
            K = lab.shape[1]
            ph, pw = lab.shape[-2], lab.shape[-1]
            lout = torch.empty_like(lab)
            engine._chk(lib.rd_augment_flip_labels(p_(lab), p_(lout), B, K, ph, pw, p_(boxes) if (boxes is not None and self.do_random_horizontal_flip) else None,
                                                   p_(pd), float(W), st), "rd_augment_flip_labels")
            if boxes is not None and self.do_random_vertical_flip:
                engine._chk(lib.rd_augment_vflip_boxes(p_(boxes), B, boxes.shape[1], p_(pd), float(H), st), "rd_augment_vflip_boxes")
            outputs.append([lout])
        elif boxes is not None and flips:
            raise NotImplementedError("boxes are flipped together with the ground-truth crops")
        if len(points_arr) > 0:
            pts_out = []
            for i, pt in enumerate(points_arr):      # radar points get the noise and are NOT flipped (rcnet_transforms.py:157-166, :174-217)
                nz = noise[i] if self.do_random_noise else None
                if nz is not None:
                    src = pt if (pt.dtype == torch.float32 and pt.is_contiguous()) else pt.float().contiguous()
                    nd = nz.to(dev, non_blocking=True)
                    res = torch.empty_like(src)
                    engine._chk(lib.rd_add(p_(src), p_(nd), p_(res), src.numel(), engine.RD_F32, st), "rd_add")
                    pt = res
                pts_out.append(pt)
            outputs.append(pts_out)
        if boxes is not None:
            outputs.append([boxes])
        return outputs[0] if len(outputs) == 1 else outputs


def crop_patches(ground_truth_padded, radar_points, patch_size):
    """data/datasets.py:254-272 on the device: (B,1,Hp,Wp) zero-padded dense depth, (B,K,3) points in padded coordinates ->
    (B,K,1,ph,pw) crops [int(y) - ph/2 : int(y) + ph/2, int(x) - pw/2 : int(x) + pw/2]."""
    B, K = radar_points.shape[0], radar_points.shape[1]
    ph, pw = int(patch_size[0]), int(patch_size[1])
    gt = ground_truth_padded if (ground_truth_padded.dtype == torch.float32 and ground_truth_padded.is_contiguous()) else ground_truth_padded.float().contiguous()
    pts = radar_points if (radar_points.dtype == torch.float32 and radar_points.is_contiguous()) else radar_points.float().contiguous()
    crops = torch.empty((B, K, 1, ph, pw), dtype=torch.float32, device=gt.device)
    engine._chk(engine.L().rd_crop_patches(engine._p(gt), engine._p(pts), engine._p(crops), B, K, gt.shape[-2], gt.shape[-1], ph, pw, engine._stream(gt)),
                "rd_crop_patches")
    return crops
