"""ORACLE (test infrastructure only): PyTorch-CPU nn.Module restatement of tf_efficientnet_lite3's feature extractor with
geffnet's attribute names (conv_stem, bn1, act1, blocks[i][j].{conv_dw,bn1,conv_pw,bn2 | conv_pw,bn1,conv_dw,bn2,conv_pwl,bn3}).

geffnet (rwightman/gen-efficientnet-pytorch, fetched by torch.hub at modules/midas/blocks.py:45-50, unpinned) is NOT in the
reference tree: this follows its published architecture.  It doubles as the `torch.hub.load` stand-in
when tests/golden/make_golden_sml.py imports the reference.  Parity of the backbone itself is therefore unpinned.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


from .precision import Precision  # noqa: E402  (bf16-mode rounding points; identity by default)


class Conv2dSame(nn.Conv2d):
    """TF-"SAME": pad_total = max((ceil(i/s)-1)*s + k - i, 0), leading pad = total // 2."""

    def forward(self, x):
        ih, iw = x.shape[-2:]
        k, s = self.kernel_size[0], self.stride[0]
        ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
        pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
        w = Precision.w(self.weight) if self.groups == 1 else self.weight      # depthwise weights stay fp32 on the HIP path
        return Precision.r(F.conv2d(x, w, self.bias, self.stride, 0, self.dilation, self.groups))


class ReLU6R(nn.ReLU6):
    def forward(self, x):
        return Precision.r(super().forward(x))


def _bn(c):
    return nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)


class DepthwiseSeparableConv(nn.Module):
    def __init__(self, cin, cout, k, s):
        super().__init__()
        self.conv_dw = Conv2dSame(cin, cin, k, s, groups=cin, bias=False)
        self.bn1 = _bn(cin)
        self.act1 = ReLU6R()
        self.conv_pw = Conv2dSame(cin, cout, 1, bias=False)
        self.bn2 = _bn(cout)
        self.has_residual = s == 1 and cin == cout

    def forward(self, x):
        h = self.bn2(self.conv_pw(self.act1(self.bn1(self.conv_dw(x)))))
        return Precision.r(h + x if self.has_residual else h)


class InvertedResidual(nn.Module):
    def __init__(self, cin, cout, k, s, exp=6):
        super().__init__()
        mid = cin * exp
        self.conv_pw = Conv2dSame(cin, mid, 1, bias=False)
        self.bn1 = _bn(mid)
        self.act1 = ReLU6R()
        self.conv_dw = Conv2dSame(mid, mid, k, s, groups=mid, bias=False)
        self.bn2 = _bn(mid)
        self.act2 = ReLU6R()
        self.conv_pwl = Conv2dSame(mid, cout, 1, bias=False)
        self.bn3 = _bn(cout)
        self.has_residual = s == 1 and cin == cout

    def forward(self, x):
        h = self.act1(self.bn1(self.conv_pw(x)))
        h = self.act2(self.bn2(self.conv_dw(h)))
        h = self.bn3(self.conv_pwl(h))
        return Precision.r(h + x if self.has_residual else h)


def _round_channels(c, mult=1.2, div=8):
    c *= mult
    new = max(div, int(c + div / 2) // div * div)
    if new < 0.9 * c:
        new += div
    return new


ARCH = [("ds", 1, 3, 1, 16), ("ir", 2, 3, 2, 24), ("ir", 2, 5, 2, 40), ("ir", 3, 3, 2, 80), ("ir", 3, 5, 1, 112), ("ir", 4, 5, 2, 192),
        ("ir", 1, 3, 1, 320)]


class EfficientNetLite3Features(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv_stem = Conv2dSame(3, 32, 3, 2, bias=False)
        self.bn1 = _bn(32)
        self.act1 = ReLU6R()
        blocks, cin = [], 32
        for si, (typ, r, k, s, c) in enumerate(ARCH):
            cout = _round_channels(c)
            rep = r if si in (0, len(ARCH) - 1) else int(math.ceil(r * 1.4))
            stage = []
            for i in range(rep):
                stage.append((DepthwiseSeparableConv if typ == "ds" else InvertedResidual)(cin, cout, k, s if i == 0 else 1))
                cin = cout
            blocks.append(nn.Sequential(*stage))
        self.blocks = nn.Sequential(*blocks)
