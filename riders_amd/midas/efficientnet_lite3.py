"""tf_efficientnet_lite3 feature extractor (parameter containers + HIP forward) with the key names of
rwightman/gen-efficientnet-pytorch ("geffnet"), which the reference pulls through torch.hub at
modules/midas/blocks.py:44-64.  geffnet is NOT part of the reference tree and the hub call is unpinned, so this is a
restatement of the published architecture: stem conv3x3 s2 -> 32, BN, ReLU6; stages
ds_r1_k3_s1_c16, ir_r2_k3_s2_e6_c24, ir_r2_k5_s2_e6_c40, ir_r3_k3_s2_e6_c80, ir_r3_k5_s1_e6_c112, ir_r4_k5_s2_e6_c192,
ir_r1_k3_s1_e6_c320 scaled by channel x1.2 / depth x1.4 (first & last repeat fixed), no SE, ReLU6, TF-"SAME" padding,
BatchNorm eps 1e-3 / momentum 0.01.  Parity of this block is therefore "unpinned" (DESIGN.md).
"""
import math

import torch
import torch.nn as nn

from .. import engine
from ..engine import ACT_NONE, ACT_RELU6

BN_EPS, BN_MOM = 1e-3, 0.01


def _round_channels(c, mult=1.2, div=8):
    c *= mult
    new = max(div, int(c + div / 2) // div * div)
    if new < 0.9 * c:
        new += div
    return new


def same_pad(i, k, s):
    """TF-SAME leading pad and output size for one axis."""
    o = -(-i // s)
    total = max((o - 1) * s + k - i, 0)
    return total // 2, o


def _bn(c):
    return nn.BatchNorm2d(c, eps=BN_EPS, momentum=BN_MOM)


class _Counted(nn.Module):
    """BN layers bump num_batches_tracked lazily on the host (flushed when a state_dict is taken)."""

    def _setup_counter(self, bns):
        self._bns, self._pending = bns, 0
        self.register_state_dict_pre_hook(lambda m, p, k: m._flush())

    def _flush(self):
        if self._pending:
            for b in self._bns:
                getattr(self, b).num_batches_tracked += self._pending
            self._pending = 0


def _conv_same(x, conv, bn, act, training, residual=None):
    k, s = conv.kernel_size[0], conv.stride[0]
    ph, oh = same_pad(x.shape[1], k, s)
    pw, ow = same_pad(x.shape[2], k, s)
    assert ph == pw, "asymmetric SAME padding differs between H and W (odd/even mix) - not supported"
    return engine.conv_block(x, conv.weight, stride=s, pad=ph, out_hw=(oh, ow), bn=bn, act=act, residual=residual, training=training)


def _dw_same(x, conv, bn, act, training):
    k, s = conv.kernel_size[0], conv.stride[0]
    ph, oh = same_pad(x.shape[1], k, s)
    pw, ow = same_pad(x.shape[2], k, s)
    assert ph == pw
    return engine.dwconv_block(x, conv.weight, stride=s, pad=ph, out_hw=(oh, ow), bn=bn, act=act, training=training)


class DepthwiseSeparableConv(_Counted):
    def __init__(self, cin, cout, k, s):
        super().__init__()
        self.conv_dw = nn.Conv2d(cin, cin, k, s, padding=0, groups=cin, bias=False)
        self.bn1 = _bn(cin)
        self.act1 = nn.ReLU6(inplace=True)
        self.conv_pw = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn2 = _bn(cout)
        self.has_residual = s == 1 and cin == cout
        self._setup_counter(["bn1", "bn2"])

    def _fwd(self, x):
        if self.training:
            self._pending += 1
        h = _dw_same(x, self.conv_dw, self.bn1, ACT_RELU6, self.training)
        return _conv_same(h, self.conv_pw, self.bn2, ACT_NONE, self.training, residual=x if self.has_residual else None)


class InvertedResidual(_Counted):
    def __init__(self, cin, cout, k, s, exp=6):
        super().__init__()
        mid = cin * exp
        self.conv_pw = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = _bn(mid)
        self.act1 = nn.ReLU6(inplace=True)
        self.conv_dw = nn.Conv2d(mid, mid, k, s, padding=0, groups=mid, bias=False)
        self.bn2 = _bn(mid)
        self.act2 = nn.ReLU6(inplace=True)
        self.conv_pwl = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = _bn(cout)
        self.has_residual = s == 1 and cin == cout
        self._setup_counter(["bn1", "bn2", "bn3"])

    def _fwd(self, x):
        if self.training:
            self._pending += 1
        h = _conv_same(x, self.conv_pw, self.bn1, ACT_RELU6, self.training)
        h = _dw_same(h, self.conv_dw, self.bn2, ACT_RELU6, self.training)
        return _conv_same(h, self.conv_pwl, self.bn3, ACT_NONE, self.training, residual=x if self.has_residual else None)


class _Stem(_Counted):
    """conv_stem / bn1 / act1 live directly in layer1 (indices 0,1,2); this helper only runs them."""


ARCH = [  # (type, repeats, k, s, c)
    ("ds", 1, 3, 1, 16), ("ir", 2, 3, 2, 24), ("ir", 2, 5, 2, 40), ("ir", 3, 3, 2, 80),
    ("ir", 3, 5, 1, 112), ("ir", 4, 5, 2, 192), ("ir", 1, 3, 1, 320),
]


def build_blocks():
    blocks, cin = [], 32
    for si, (typ, r, k, s, c) in enumerate(ARCH):
        cout = _round_channels(c)
        rep = r if si in (0, len(ARCH) - 1) else int(math.ceil(r * 1.4))
        stage = []
        for i in range(rep):
            st = s if i == 0 else 1
            stage.append(DepthwiseSeparableConv(cin, cout, k, st) if typ == "ds" else InvertedResidual(cin, cout, k, st))
            cin = cout
        blocks.append(nn.Sequential(*stage))
    return blocks


def make_pretrained():
    """The `pretrained` module of blocks._make_efficientnet_backbone: layer1 = [conv_stem, bn1, act1, blocks0, blocks1],
    layer2 = [blocks2], layer3 = [blocks3, blocks4], layer4 = [blocks5, blocks6]."""
    blocks = build_blocks()
    pre = nn.Module()
    pre.layer1 = nn.Sequential(nn.Conv2d(3, 32, 3, 2, padding=0, bias=False), _bn(32), nn.ReLU6(inplace=True), blocks[0], blocks[1])
    pre.layer2 = nn.Sequential(blocks[2])
    pre.layer3 = nn.Sequential(blocks[3], blocks[4])
    pre.layer4 = nn.Sequential(blocks[5], blocks[6])
    for m in pre.modules():
        if isinstance(m, nn.Conv2d):
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
            nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
    return pre


def run_layer1(layer1, x, training):
    conv, bn = layer1[0], layer1[1]
    if training:
        layer1._stem_pending = getattr(layer1, "_stem_pending", 0) + 1
    h = _conv_same(x, conv, bn, ACT_RELU6, training)
    for stage in (layer1[3], layer1[4]):
        for blk in stage:
            h = blk._fwd(h)
    return h


def run_stages(layer, x):
    for stage in layer:
        for blk in stage:
            x = blk._fwd(x)
    return x
