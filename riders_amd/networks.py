"""RC-Net encoders / decoder on MI355X.  Same class names, constructor arguments, forward() signatures and
state_dict keys as the reference's RCNet/networks.py (ResNetEncoder :10, FullyConnectedEncoder :273,
RCNetEncoder :335, MultiScaleDecoder :458).

Each top-level forward is ONE autograd node (engine.run_region) that hand-schedules HIP launches; tensors
crossing the module boundary are logical NCHW with channels_last strides (physically NHWC, zero copy).
"""
import torch

from . import engine, net_utils
from .linear_attention import LocalFeatureTransformer


def boxes_to_rois(b_boxes, device=None):
    """list of B tensors (K,4) (x1,y1,x2,y2), or one (B,K,4) tensor -> (R,5) fp32 rows (batch index, x1, y1, x2, y2), image-major.
    Mirrors torchvision.ops._utils.convert_boxes_to_roi_format used by roi_pool (networks.py:418-433); one rd_boxes_to_rois launch
    (per image for a list) instead of a fill + concatenation per image.  A ready (R,5) tensor is passed through."""
    if torch.is_tensor(b_boxes):
        if b_boxes.dim() == 2 and b_boxes.shape[1] == 5:
            return b_boxes.to(torch.float32).contiguous()
        if b_boxes.dim() == 2 and b_boxes.shape[1] == 4:
            b_boxes = b_boxes[None]
        b_boxes = [b_boxes]
        chunks = [(b_boxes[0], 0)]
    else:
        chunks, img = [], 0
        for b in b_boxes:
            chunks.append((b[None], img))
            img += 1
    total = sum(c.shape[0] * c.shape[1] for c, _ in chunks)
    dev = chunks[0][0].device
    rois = torch.empty((total, 5), dtype=torch.float32, device=dev)
    lib, row = engine.L(), 0
    for c, first in chunks:
        c = c if (c.dtype == torch.float32 and c.is_contiguous()) else c.to(torch.float32).contiguous()
        B, K = c.shape[0], c.shape[1]
        if B * K:
            engine._chk(lib.rd_boxes_to_rois(engine._p(c), engine._p(rois[row:]), B, K, first, engine._stream(c)), "rd_boxes_to_rois")
        row += B * K
    return rois


class ResNetEncoder(torch.nn.Module):
    """ResNet-18/34 style encoder with skip connections.  Reference: RCNet/networks.py:10-270."""

    def __init__(self, n_layer, input_channels=3, n_filters=[32, 64, 128, 256, 256],
                 weight_initializer='kaiming_uniform', activation_func='leaky_relu', use_batch_norm=False):
        super(ResNetEncoder, self).__init__()
        if n_layer == 18:
            n_blocks = [2, 2, 2, 2]
        elif n_layer == 34:
            n_blocks = [3, 4, 6, 3]
        else:
            raise ValueError('Only supports 18, 34 layer architecture')
        resnet_block = net_utils.ResNetBlock
        for n in range(len(n_filters) - len(n_blocks) - 1):
            n_blocks = n_blocks + [n_blocks[-1]]
        network_depth = len(n_filters)
        assert network_depth < 8, 'Does not support network depth of 8 or more'
        assert network_depth == len(n_blocks) + 1
        activation_func = net_utils.activation_func(activation_func)

        kw = dict(weight_initializer=weight_initializer, activation_func=activation_func, use_batch_norm=use_batch_norm)
        self.conv1 = net_utils.Conv2d(input_channels, n_filters[0], kernel_size=7, stride=2, **kw)
        self.max_pool = torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.blocks2 = self._make_layer(resnet_block, n_blocks[0], n_filters[0], n_filters[1], 1, **kw)
        self.blocks3 = self._make_layer(resnet_block, n_blocks[1], n_filters[1], n_filters[2], 2, **kw)
        self.blocks4 = self._make_layer(resnet_block, n_blocks[2], n_filters[2], n_filters[3], 2, **kw)
        self.blocks5 = self._make_layer(resnet_block, n_blocks[3], n_filters[3], n_filters[4], 2, **kw)
        self.blocks6 = self._make_layer(resnet_block, n_blocks[4], n_filters[4], n_filters[5], 2, **kw) \
            if network_depth > 5 else None
        self.blocks7 = self._make_layer(resnet_block, n_blocks[5], n_filters[5], n_filters[6], 2, **kw) \
            if network_depth > 6 else None

    def _make_layer(self, network_block, n_block, in_channels, out_channels, stride, weight_initializer,
                    activation_func, use_batch_norm):
        blocks = []
        for n in range(n_block):
            if n != 0:
                in_channels, stride = out_channels, 1
            blocks.append(network_block(in_channels=in_channels, out_channels=out_channels, stride=stride,
                                        weight_initializer=weight_initializer, activation_func=activation_func,
                                        use_batch_norm=use_batch_norm))
        return torch.nn.Sequential(*blocks)

    def _fwd(self, x):
        """x (N,H,W,C) -> latent, [skips] (all NHWC)."""
        layers = [x]
        layers.append(self.conv1._fwd(layers[-1]))
        mp = self.max_pool
        h = engine.maxpool(layers[-1], int(mp.kernel_size), int(mp.stride), int(mp.padding))
        for stage in (self.blocks2, self.blocks3, self.blocks4, self.blocks5, self.blocks6, self.blocks7):
            if stage is None:
                continue
            for blk in stage:
                h = blk._fwd(h)
            layers.append(h)
        return layers[-1], layers[1:-1]

    def forward(self, x):
        def run(x):
            latent, skips = self._fwd(engine.from_nchw(x))
            return tuple(engine.to_nchw_out(t, x.dtype) for t in [latent] + list(skips))
        outs = engine.run_region(run, (x,), list(self.parameters()))
        return outs[0], list(outs[1:])


class FullyConnectedEncoder(torch.nn.Module):
    """Radar-point MLP.  Reference: RCNet/networks.py:273-332 (activation also after the last layer)."""

    def __init__(self, input_channels=3, n_neurons=[32, 64, 96, 128, 256], latent_size=29 * 10,
                 weight_initializer='kaiming_uniform', activation_func='leaky_relu'):
        super(FullyConnectedEncoder, self).__init__()
        activation_func = net_utils.activation_func(activation_func)
        dims = [input_channels] + list(n_neurons[:5]) + [latent_size]
        self.mlp = torch.nn.Sequential(*[
            net_utils.FullyConnected(in_features=dims[i], out_features=dims[i + 1],
                                     weight_initializer=weight_initializer, activation_func=activation_func)
            for i in range(6)])

    def _fwd(self, x):
        for fc in self.mlp:
            x = fc._fwd(x)
        return x

    def forward(self, x):
        def run(x):
            return self._fwd(x if x.is_contiguous() else x.contiguous())
        return engine.run_region(run, (x,), list(self.parameters()))


class RCNetEncoder(torch.nn.Module):
    """Image encoder -> ROI max-pool per radar point -> point MLP -> linear cross-attention -> concat.
    Reference: RCNet/networks.py:335-451."""

    def __init__(self, input_channels_image=3, input_channels_depth=3, input_patch_size_image=(900, 288),
                 n_filters_encoder_image=[32, 64, 128, 128, 128], n_neurons_encoder_depth=[32, 64, 128, 128, 128],
                 latent_size_depth=128 * 29 * 10, weight_initializer='kaiming_uniform', activation_func='leaky_relu',
                 use_batch_norm=False):
        super(RCNetEncoder, self).__init__()
        self.n_neuron_latent_depth = n_neurons_encoder_depth[-1]
        self.encoder_image = ResNetEncoder(n_layer=18, input_channels=input_channels_image,
                                           n_filters=n_filters_encoder_image, weight_initializer=weight_initializer,
                                           activation_func=activation_func, use_batch_norm=use_batch_norm)
        self.attention = LocalFeatureTransformer(['self', 'cross'], n_layers=4, d_model=self.n_neuron_latent_depth)
        self.encoder_depth = FullyConnectedEncoder(input_channels=input_channels_depth,
                                                   n_neurons=n_neurons_encoder_depth, latent_size=latent_size_depth,
                                                   weight_initializer=weight_initializer,
                                                   activation_func=activation_func)
        self.input_patch_size_image = input_patch_size_image

    def _fwd(self, image, points, rois):
        """image (B,H,W,C) act dtype, points (R,X) fp32, rois (R,5) fp32 -> latent (R,lh,lw,2C), [pooled skips]."""
        shape = self.input_patch_size_image
        latent_height = int(shape[-2] // 32.0)
        latent_width = int(shape[-1] // 32.0)
        skip_scales = [1 / 2.0, 1 / 4.0, 1 / 8.0, 1 / 16.0, 1 / 32.0, 1 / 64.0, 1 / 128.0]
        skip_feature_sizes = [(int(shape[-2] * s), int(shape[-1] * s)) for s in skip_scales]
        C = self.n_neuron_latent_depth
        R = points.shape[0]
        L = latent_height * latent_width

        latent_image, skips_image = self.encoder_image._fwd(image)
        engine.tap("enc.latent_image", latent_image)
        for i, s_ in enumerate(skips_image):
            engine.tap("enc.skip%d" % i, s_)
        engine.stage_mark("attention_done")   # backward: RoI pooling, point MLP and transformer gradients are final here
        skips_image_pooled = [engine.roi_pool(skips_image[i], rois, skip_feature_sizes[i], skip_scales[i])
                              for i in range(len(skips_image))]
        # the transformer keeps both token streams in ONE (2 R L, C) matrix: the pooled latent and the cast point-MLP tokens are written
        # straight into its two halves (engine.rows_cat then has nothing to copy)
        merged = engine.fused_loftr() and latent_image.shape[3] == C and latent_image.dtype == engine.act_dtype()
        tok = torch.empty((2 * R * L, C), dtype=latent_image.dtype, device=latent_image.device) if merged else None
        latent_image_pooled = engine.roi_pool(latent_image, rois, (latent_height, latent_width), 1 / 32.0,
                                              out=None if tok is None else tok[R * L:].view(R, latent_height, latent_width, C))
        engine.tap("enc.latent_pooled", latent_image_pooled)
        for i, s_ in enumerate(skips_image_pooled):
            engine.tap("enc.skip%d_pooled" % i, s_)

        # point MLP stays in fp32 (raw pixel coordinates in the hundreds), output viewed (R, C, L) -> tokens (R, L, C)
        latent_depth = self.encoder_depth._fwd(points)
        engine.tap("enc.mlp_out", latent_depth)
        tokens_depth = engine.transpose_last2(latent_depth, R, C, L)          # (R, L, C)
        tokens_depth = engine.alias(tokens_depth, tokens_depth.view(R * L, C))
        tokens_depth = engine.input_cast(tokens_depth, out=None if tok is None else tok[:R * L])
        tokens_image = engine.alias(latent_image_pooled, latent_image_pooled.view(R * L, C))

        depth_tf, image_tf = self.attention._fwd(tokens_depth, tokens_image, R, L, L)
        latent = engine.concat_channels(image_tf, depth_tf)                   # cat([image_tf, depth_tf], dim=1)
        engine.tap("enc.latent", latent)
        latent = engine.alias(latent, latent.view(R, latent_height, latent_width, 2 * C))
        return latent, skips_image_pooled

    def forward(self, image, points, b_boxes):
        rois = boxes_to_rois(b_boxes)

        def run(image, points):
            pts = points if points.is_contiguous() else points.contiguous()
            latent, skips = self._fwd(engine.from_nchw(image), pts, rois)
            return tuple(engine.to_nchw_out(t, image.dtype) for t in [latent] + list(skips))
        outs = engine.run_region(run, (image, points), list(self.parameters()))
        return outs[0], list(outs[1:])


class MultiScaleDecoder(torch.nn.Module):
    """Multi-scale decoder with skip connections (n_resolution = 1, deconv_type 'up', linear output as configured
    at RCNet/rcnet_model.py:84-94).  Reference: RCNet/networks.py:458-779."""

    def __init__(self, input_channels=256, output_channels=1, n_resolution=1, n_filters=[256, 128, 64, 32, 16],
                 n_skips=[256, 128, 64, 32, 0], weight_initializer='kaiming_uniform', activation_func='leaky_relu',
                 output_func='linear', use_batch_norm=False, deconv_type='up'):
        super(MultiScaleDecoder, self).__init__()
        network_depth = len(n_filters)
        assert network_depth < 8, 'Does not support network depth of 8 or more'
        assert n_resolution > 0 and n_resolution < network_depth
        if n_resolution != 1 or 'upsample' in output_func:
            raise NotImplementedError('only n_resolution=1 without upsampled outputs is on the RIDERS hot path')
        if network_depth != 5:
            raise NotImplementedError('RIDERS uses a 5-level decoder (rcnet_model.py:77-94)')
        self.n_resolution = n_resolution
        self.output_func = output_func
        act = net_utils.activation_func(activation_func)
        out_act = net_utils.activation_func(output_func)
        kw = dict(weight_initializer=weight_initializer, activation_func=act, use_batch_norm=use_batch_norm,
                  deconv_type=deconv_type)
        self.deconv6 = None
        self.deconv5 = None
        chans = [input_channels] + list(n_filters)
        for i, name in enumerate(['deconv4', 'deconv3', 'deconv2', 'deconv1', 'deconv0']):
            setattr(self, name, net_utils.DecoderBlock(chans[i], n_skips[i], n_filters[i], **kw))
        self.output0 = net_utils.Conv2d(n_filters[-1], output_channels, kernel_size=3, stride=1,
                                        weight_initializer=weight_initializer, activation_func=out_act,
                                        use_batch_norm=False)

    def _fwd(self, x, skips, shape=None):
        # every block's output is consumed by the next block's up-convolution (or the output convolution): virtual (engine.LazyAct) at lazy_bn level 2
        n = len(skips) - 1
        h = engine.tap("dec.deconv4", self.deconv4._fwd(x, skips[n], lazy=2)); n -= 1
        h = engine.tap("dec.deconv3", self.deconv3._fwd(h, skips[n], lazy=2)); n -= 1
        h = engine.tap("dec.deconv2", self.deconv2._fwd(h, skips[n], lazy=2)); n -= 1
        h = engine.tap("dec.deconv1", self.deconv1._fwd(h, skips[n], lazy=2)); n -= 1
        # deconv0's output feeds the one-channel output convolution only: virtual at level 1 when the fused head kernels take the pair (engine bn_head)
        lz0 = 1 if engine.head_route(self.output0.conv.in_channels) else 2
        if n == 0:
            h = self.deconv0._fwd(h, skips[n], lazy=lz0)
        else:
            h = self.deconv0._fwd(h, shape=tuple(shape[-2:]), lazy=lz0)
        return [self.output0._fwd(h)]

    def forward(self, x, skips, shape=None):
        ns = len(skips)

        def run(x, *sk):
            outs = self._fwd(engine.from_nchw(x), [engine.from_nchw(s) for s in sk], shape)
            return tuple(engine.to_nchw_out(o, x.dtype) for o in outs)
        outs = engine.run_region(run, (x,) + tuple(skips), list(self.parameters()))
        return list(outs)
