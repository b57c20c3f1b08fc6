"""MidasNet_small_videpth (Scale Map Learner) on MI355X; same constructor / forward(x, d) / state_dict keys as
modules/midas/midas_net_custom.py:22-133.  The whole forward is one engine region."""
import math

import torch
import torch.nn as nn

from .. import engine
from ..engine import ACT_RELU
from . import efficientnet_lite3
from .base_model import BaseModel
from .blocks import FeatureFusionBlock_custom, OutputConv, _make_encoder


def weights_init(m):
    if isinstance(m, nn.Conv2d):
        n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
        m.weight.data.normal_(0, math.sqrt(2.0 / n))
        if m.bias is not None:
            m.bias.data.zero_()
    elif isinstance(m, nn.BatchNorm2d):
        m.weight.data.fill_(1)
        m.bias.data.zero_()


class MidasNet_small_videpth(BaseModel):
    def __init__(self, device='cpu', path=None, features=64, backbone="efficientnet_lite3", non_negative=False, exportable=True,
                 channels_last=False, align_corners=True, blocks={'expand': True}, in_channels=2, regress='r', min_pred=None,
                 max_pred=None):
        print("Loading weights: ", path)
        super(MidasNet_small_videpth, self).__init__()
        use_pretrained = False if path else True
        self.channels_last = channels_last
        self.blocks = blocks
        self.backbone = backbone
        self.groups = 1
        self.regress = regress
        self.min_pred = min_pred
        self.max_pred = max_pred
        self.expand = bool(self.blocks.get('expand', False)) if isinstance(self.blocks, dict) else False
        f1, f2, f3, f4 = (features, features * 2, features * 4, features * 8) if self.expand else (features,) * 4
        self.first = nn.Sequential(nn.Conv2d(in_channels, 3, kernel_size=3, stride=1, padding=1), nn.BatchNorm2d(3), nn.ReLU(inplace=True))
        self.first.apply(weights_init)
        self.pretrained, self.scratch = _make_encoder(self.backbone, features, use_pretrained, groups=self.groups, expand=self.expand,
                                                      exportable=exportable)
        self.scratch.activation = nn.ReLU(False)
        self.scratch.refinenet4 = FeatureFusionBlock_custom(f4, self.scratch.activation, deconv=False, bn=False, expand=self.expand, align_corners=align_corners)
        self.scratch.refinenet3 = FeatureFusionBlock_custom(f3, self.scratch.activation, deconv=False, bn=False, expand=self.expand, align_corners=align_corners)
        self.scratch.refinenet2 = FeatureFusionBlock_custom(f2, self.scratch.activation, deconv=False, bn=False, expand=self.expand, align_corners=align_corners)
        self.scratch.refinenet1 = FeatureFusionBlock_custom(f1, self.scratch.activation, deconv=False, bn=False, align_corners=align_corners)
        self.scratch.output_conv = OutputConv(features, self.groups, self.scratch.activation, non_negative)
        self._first_pending = 0
        self.register_state_dict_pre_hook(lambda m, p, k: m._flush_counters())
        if path:
            self.load(path)
        self.to(device)

    def _flush_counters(self):
        if self._first_pending:
            self.first[1].num_batches_tracked += self._first_pending
            self._first_pending = 0
        sp = getattr(self.pretrained.layer1, "_stem_pending", 0)
        if sp:
            self.pretrained.layer1[1].num_batches_tracked += sp
            self.pretrained.layer1._stem_pending = 0

    def _fwd(self, x, d):
        """x (B,H,W,C) activation dtype, d (B,H,W,1) fp32 -> pred (B,H,W,1) fp32."""
        tr = self.training
        if tr:
            self._first_pending += 1
        c, bn = self.first[0], self.first[1]
        layer_0 = engine.conv_block(x, c.weight, bias=c.bias, bn=bn, act=ACT_RELU, training=tr)
        layer_1 = efficientnet_lite3.run_layer1(self.pretrained.layer1, layer_0, tr)
        layer_2 = efficientnet_lite3.run_stages(self.pretrained.layer2, layer_1)
        layer_3 = efficientnet_lite3.run_stages(self.pretrained.layer3, layer_2)
        engine.stage_mark("layer4_done")      # backward-order boundaries for the bucketed gradient all-reduce (parallel.sml_stages)
        layer_4 = efficientnet_lite3.run_stages(self.pretrained.layer4, layer_3)
        engine.stage_mark("scratch_done")
        s = self.scratch
        l1 = engine.conv_block(layer_1, s.layer1_rn.weight)
        l2 = engine.conv_block(layer_2, s.layer2_rn.weight)
        l3 = engine.conv_block(layer_3, s.layer3_rn.weight)
        l4 = engine.conv_block(layer_4, s.layer4_rn.weight)
        path_4 = s.refinenet4._fwd(l4)
        path_3 = s.refinenet3._fwd(path_4, l3)
        path_2 = s.refinenet2._fwd(path_3, l2)
        path_1 = s.refinenet1._fwd(path_2, l1)
        out = s.output_conv._fwd(path_1)
        return engine.sml_head(out, d, self.min_pred, self.max_pred)

    def forward(self, x, d):
        dd = d if d.is_contiguous() else d.contiguous()   # (B,1,H,W): C == 1, NCHW == NHWC in memory

        def run(x, dd):
            pred = self._fwd(engine.from_nchw(x), dd)
            return engine.alias(pred, pred.view(dd.shape))
        # (d is a region INPUT: a captured region -- engine.set_autograph -- reads it from a static tensor)
        return engine.run_region(run, (x, dd), list(self.parameters()), graph_key=("MidasNet_small_videpth.forward", id(self), self.training),
                                 on_replay=self._bn_replay)

    def _bn_replay(self):
        """host-side bookkeeping of one replayed forward: the num_batches_tracked counters this module's _fwd bumps"""
        from .efficientnet_lite3 import _Counted
        if self.training:
            for m in self.modules():
                if isinstance(m, _Counted):
                    m._pending += 1
            self._first_pending += 1
            self.pretrained.layer1._stem_pending = getattr(self.pretrained.layer1, "_stem_pending", 0) + 1


class MidasNet_small_depth(BaseModel):
    """modules/midas/midas_net_custom.py:136-261, model_type 'midas-small-depth' (train_zju.py:180): an alternative head the ZJU configuration
    never selects (out of scope).  The name exists because train_zju.py:14 / val_zju.py:17 import it next to
    MidasNet_small_videpth; constructing it says so."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("MidasNet_small_depth ('midas-small-depth') is not on the RIDERS ZJU path; use MidasNet_small_videpth")
