"""MiDaS-small decoder blocks on MI355X; same names / signatures / state_dict keys as modules/midas/blocks.py
(_make_encoder :4, _make_scratch :15, ResidualConvUnit_custom :67, FeatureFusionBlock_custom :125, OutputConv :177)."""
import torch
import torch.nn as nn

from .. import engine
from ..engine import ACT_NONE, ACT_RELU
from . import efficientnet_lite3


def _make_encoder(backbone, features, use_pretrained, groups=1, expand=False, exportable=True):
    if backbone != "efficientnet_lite3":
        raise NotImplementedError("Backbone '%s' not implemented" % backbone)
    if use_pretrained:
        print("riders_amd: no network on this system - tf_efficientnet_lite3 starts from random weights; "
              "load a state_dict (same keys as the hub model) for the ImageNet initialisation")
    pretrained = efficientnet_lite3.make_pretrained()
    scratch = _make_scratch([32, 48, 136, 384], features, groups=groups, expand=expand)
    return pretrained, scratch


def _make_scratch(in_shape, out_shape, groups=1, expand=False):
    scratch = nn.Module()
    o = [out_shape, out_shape * 2, out_shape * 4, out_shape * 8] if expand else [out_shape] * 4
    for i in range(4):
        setattr(scratch, "layer%d_rn" % (i + 1), nn.Conv2d(in_shape[i], o[i], kernel_size=3, stride=1, padding=1, bias=False, groups=groups))
    return scratch


class ResidualConvUnit_custom(nn.Module):
    def __init__(self, features, activation, bn):
        super().__init__()
        if bn:
            raise NotImplementedError("bn=True is not used by MidasNet_small_videpth (midas_net_custom.py:73-76)")
        self.bn = bn
        self.groups = 1
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True, groups=self.groups)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True, groups=self.groups)
        self.activation = activation
        self.skip_add = nn.Identity()  # FloatFunctional.add == '+'; no parameters either way

    def _fwd(self, x):
        out = engine.activation(x, ACT_RELU)
        out = engine.conv_block(out, self.conv1.weight, bias=self.conv1.bias, act=ACT_RELU)       # conv1 -> activation
        return engine.conv_block(out, self.conv2.weight, bias=self.conv2.bias, residual=x)         # conv2 + x


class FeatureFusionBlock_custom(nn.Module):
    def __init__(self, features, activation, deconv=False, bn=False, expand=False, align_corners=True):
        super(FeatureFusionBlock_custom, self).__init__()
        self.deconv = deconv
        self.align_corners = align_corners
        self.groups = 1
        self.expand = expand
        out_features = features // 2 if expand else features
        self.out_conv = nn.Conv2d(features, out_features, kernel_size=1, stride=1, padding=0, bias=True, groups=1)
        self.resConfUnit1 = ResidualConvUnit_custom(features, activation, bn)
        self.resConfUnit2 = ResidualConvUnit_custom(features, activation, bn)
        self.skip_add = nn.Identity()

    def _fwd(self, *xs):
        output = xs[0]
        if len(xs) == 2:
            res = self.resConfUnit1._fwd(xs[1])
            output = engine.add_act(output, res, ACT_NONE)
        output = self.resConfUnit2._fwd(output)
        if engine.switch("fusion_conv_first"):
            # modules/midas/blocks.py:168-172 up-samples (bilinear x2) and then applies the 1x1 out_conv.  Both are linear and the interpolation
            # weights of a pixel sum to 1, so they commute exactly (bias included): the convolution runs on a quarter of the pixels and the
            # interpolation on out_features (= features / 2 when expand) channels.  Same function; in bf16 the rounding points move (the
            # convolution's output is rounded before the interpolation instead of after).
            output = engine.conv_block(output, self.out_conv.weight, bias=self.out_conv.bias, pad=0)
            return engine.bilinear2x(output, self.align_corners)
        output = engine.bilinear2x(output, self.align_corners)
        return engine.conv_block(output, self.out_conv.weight, bias=self.out_conv.bias, pad=0)


class OutputConv(nn.Module):
    def __init__(self, features, groups, activation, non_negative):
        super(OutputConv, self).__init__()
        self.output_conv = nn.Sequential(
            nn.Conv2d(features, features // 2, kernel_size=3, stride=1, padding=1, groups=groups),
            nn.Upsample(scale_factor=2, mode="bilinear"),
            nn.Conv2d(features // 2, 32, kernel_size=3, stride=1, padding=1),
            activation,
            nn.Conv2d(32, 1, kernel_size=1, stride=1, padding=0),
            nn.ReLU(True) if non_negative else nn.Identity(),
            nn.Identity(),
        )
        self.non_negative = non_negative

    def _fwd(self, x):
        c0, c2, c4 = self.output_conv[0], self.output_conv[2], self.output_conv[4]
        h = engine.conv_block(x, c0.weight, bias=c0.bias)
        h = engine.bilinear2x(h, False)
        h = engine.conv_block(h, c2.weight, bias=c2.bias, act=ACT_RELU)
        return engine.conv_block(h, c4.weight, bias=c4.bias, pad=0, act=ACT_RELU if self.non_negative else ACT_NONE)
