"""Hand-derived known-answer tests for the THIRD-PARTY arithmetic the reference calls but does not contain, so that the oracle's
restatements -- and the fixtures generated with them standing in (g13: torchvision stub, S1 / g8: cv2 stub, g9: geffnet stand-in) -- are
pinned to something other than themselves (VERDICT r02 "parity holes" (c), (d)).  Literals live in tests/golden/kat_third_party.json.

Derivations (integers, factors chosen as dyadic rationals so every product is exact in float32 and the truncation is unambiguous):
* torchvision 0.14 functional_tensor, integer images (RCNet/rcnet_transforms.py:287-346 casts to int first):
    _blend(a, b, r) = trunc(clamp(r a + (1 - r) b, 0, 255));   gray = trunc(0.2989 R + 0.587 G + 0.114 B)
    pixel A = (100,150,200): 29.89 + 88.05 + 22.8 = 140.74 -> 140;  pixel B = (10,20,30): 2.989 + 11.74 + 3.42 = 18.149 -> 18
    brightness r: trunc(min(r c, 255)): 1.25 A = (125, 187.5, 250), 1.25 B = (12.5, 25, 37.5); 0.75 A = (75, 112.5, 150), 0.75 B = (7.5, 15, 22.5);
                  1.5 A = (150, 225, 300 -> 255)
    contrast r: mean gray over the image = (140 + 18) / 2 = 79;  r c + (1 - r) 79:  0.75: A (94.75, 132.25, 169.75), B (27.25, 34.75, 42.25);
                  1.25: A (105.25, 167.75, 230.25), B (-7.25 -> 0, 5.25, 17.75)
    saturation r: r c + (1 - r) gray(pixel): 0.5: A (120, 145, 170), B (14, 19, 24);  1.25: A (125-35, 187.5-35, 250-35) = (90, 152.5, 215),
                  B (12.5-4.5, 25-4.5, 37.5-4.5) = (8, 20.5, 33)
* cv2.resize INTER_NEAREST: src index = min(floor(dst index * src / dst), src - 1).
* TF-"SAME" (geffnet Conv2dSame): out = ceil(i / s), total = max((out - 1) s + k - i, 0), before = total // 2, after = total - before.
    all-ones 4x4 input, all-ones k3 s2 kernel, pads (0, 1): windows rows/cols {0,1,2} and {2,3,(4)} -> [[9, 6], [6, 4]]
    k5 s2 on 4x4: total = 2 + 5 - 4 = 3 -> pads (1, 2): windows {(-1),0,1,2,3} = 4 valid and {1,2,3,(4),(5)} = 3 valid per axis
    -> [[16, 12], [12, 9]]
    5x5 k3 s2: total = 4 + 3 - 5 = 2 -> pads (1, 1): windows {(-1),0,1}, {1,2,3}, {3,4,(5)} -> 2,3,2 valid per axis -> outer product.
* tf_efficientnet_lite3 (published): stem 32; stages (repeats, k, s, channels) after x1.2 width / x1.4 depth scaling with the first and
    last stage's repeats fixed; features at strides 4/8/16/32 with 32/48/136/384 channels; 6,422,016 backbone parameters (8.2 M model minus
    conv_head 384 -> 1280 and classifier) -- SURVEY.md Appendix B."""
import json
import math
import os

import numpy as np
import pytest
import torch

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_third_party.json")))


def _img(px):
    return torch.tensor(px, dtype=torch.int32).t().reshape(3, 1, len(px))      # (3, 1, n pixels)


def _px(img):
    return img.reshape(3, -1).t().tolist()


def test_photometric_oracle_matches_hand_derived_integers():
    from oracle import transforms as OT
    k = KAT["photometric_int"]
    img = _img(k["image_2px"])
    assert OT.rgb_to_grayscale(img).reshape(-1).tolist() == k["gray"]
    for name, fn in (("brightness", OT.adjust_brightness), ("contrast", OT.adjust_contrast), ("saturation", OT.adjust_saturation)):
        for key, want in k.items():
            if key.startswith(name + "_"):
                got = _px(fn(img, float(key.split("_")[1])))
                assert got == want, (key, got, want)


def test_inter_nearest_index_rule():
    """oracle.sml.nearest_resize (the cv2.resize stand-in of fixtures g8 / S1) and the product's host twin sml_main.nearest_resize."""
    from oracle import sml as OS
    from riders_amd import sml_main
    for key, want in KAT["inter_nearest"].items():
        a, b = [int(v) for v in key.split("->")]
        if isinstance(want, str):
            want = [min(j * a // b, a - 1) for j in range(b)]
            assert want[-1] == 255
        src = np.arange(a, dtype=np.float32)
        assert OS.nearest_resize(np.tile(src[:, None], (1, a)), b, b)[:, 0].astype(int).tolist() == want      # rows
        assert OS.nearest_resize(np.tile(src[None, :], (a, 1)), b, b)[0].astype(int).tolist() == want         # columns
        t4 = torch.arange(a, dtype=torch.float32).reshape(1, 1, a, 1).expand(1, 1, a, a).contiguous()
        assert sml_main.nearest_resize(t4, b, b)[0, 0, :, 0].int().tolist() == want


def test_tf_same_padding_rule_and_oracle_conv():
    from oracle.effnet_lite3_torch import Conv2dSame
    from riders_amd.midas.efficientnet_lite3 import same_pad
    for i, k, s, before, after, out in KAT["tf_same"]["cases_i_k_s_before_after_out"]:
        lead, o = same_pad(i, k, s)
        total = max((math.ceil(i / s) - 1) * s + k - i, 0)
        assert (lead, total - lead, o) == (before, after, out), (i, k, s)
    for key, hw, k in (("ones_4x4_k3_s2", 4, 3), ("ones_4x4_k5_s2", 4, 5), ("ones_5x5_k3_s2", 5, 3)):
        conv = Conv2dSame(1, 1, k, 2, bias=False)
        with torch.no_grad():
            conv.weight.fill_(1.0)
            got = conv(torch.ones(1, 1, hw, hw))[0, 0].int().tolist()
        assert got == KAT["tf_same"][key], (key, got)


def test_effnet_lite3_invariants():
    """S3: the backbone is third-party (geffnet via torch.hub, absent here): pin what IS published -- parameter count, key count and names
    shape, stage table, feature channels / strides -- on both the oracle's restatement and the product's module."""
    import contextlib
    import io
    from oracle.effnet_lite3_torch import EfficientNetLite3Features
    from riders_amd import sml_main
    k = KAT["effnet_lite3"]
    o = EfficientNetLite3Features()
    assert sum(p.numel() for p in o.parameters()) == k["params"] and len(o.state_dict()) == k["state_dict_keys"]
    with contextlib.redirect_stdout(io.StringIO()):
        m = sml_main.build_model(torch.device("cpu"))
    sd = m.state_dict()
    pk = [n for n in sd if n.startswith("pretrained.")]
    assert len(pk) == k["state_dict_keys"]
    assert sum(sd[n].numel() for n in pk if "running" not in n and "num_batches" not in n) == k["params"]
    # geffnet key names: stem, bn1, then blocks as layerL.S.N.{conv_pw,bn1,conv_dw,bn2,conv_pwl,bn3} (DS block: conv_dw,bn1,conv_pw,bn2)
    assert "pretrained.layer1.0.weight" in sd and tuple(sd["pretrained.layer1.0.weight"].shape) == (k["stem_cout"], 3, 3, 3)
    assert "pretrained.layer1.3.0.conv_dw.weight" in sd and "pretrained.layer1.3.0.conv_pw.weight" in sd
    stage_prefix = ["pretrained.layer1.3", "pretrained.layer1.4", "pretrained.layer2.0", "pretrained.layer3.0", "pretrained.layer3.1",
                    "pretrained.layer4.0", "pretrained.layer4.1"]
    for (rep, kk, s, cout), pre in zip(k["stages_repeats_k_s_cout"], stage_prefix):
        blocks = sorted({n[len(pre) + 1:].split(".")[0] for n in sd if n.startswith(pre + ".")}, key=int)
        assert len(blocks) == rep, (pre, blocks)
        dw = sd["%s.0.conv_dw.weight" % pre]
        assert dw.shape[-1] == kk and dw.shape[1] == 1, pre
        last = "%s.%d.%s.weight" % (pre, rep - 1, "conv_pw" if pre.endswith("layer1.3") else "conv_pwl")
        assert sd[last].shape[0] == cout, (pre, tuple(sd[last].shape))
    # oracle restatement of the same table: feature maps at strides 4/8/16/32
    feats = []
    x = torch.zeros(1, 3, 64, 96)
    with torch.no_grad():
        h = o.act1(o.bn1(o.conv_stem(x)))
        for i, blk in enumerate(o.blocks):
            h = blk(h)
            if i in (1, 2, 4, 6):      # stages after which layer1 / 2 / 3 / 4 end (blocks.py:54-64 slices them 0:2, 2:3, 3:5, 5:9)
                feats.append(h)
    assert [f.shape[1] for f in feats] == k["feature_channels"]
    assert [64 // f.shape[2] for f in feats] == k["feature_strides"] and [96 // f.shape[3] for f in feats] == k["feature_strides"]


def _photometric_on_device(dev):
    """The same integer KATs through the HIP augmentation kernels (rd_augment_gray_partials / rd_augment_image)."""
    from riders_amd import engine
    from riders_amd.rcnet_transforms import Transforms
    k = KAT["photometric_int"]
    engine.set_compute_dtype("fp32")
    img = _img(k["image_2px"]).float()[None].to(dev)          # (1, 3, 1, 2)
    tr = Transforms(normalized_image_range=[0, 255], random_brightness=[0.5, 1.5], random_contrast=[0.5, 1.5], random_saturation=[0.5, 1.5])
    for col, name in ((0, "brightness"), (2, "contrast"), (4, "saturation")):
        for key, want in k.items():
            if not key.startswith(name + "_"):
                continue
            p = torch.zeros((1, 8), dtype=torch.float32)
            p[0, col], p[0, col + 1] = 1.0, float(key.split("_")[1])
            out = tr.transform([img], params=p)[0]
            got = out[0].float().cpu().reshape(3, -1).t().int().tolist()
            assert got == want, (key, got, want)


def test_photometric_kernels_match_hand_derived_integers_emu(emu):
    _photometric_on_device(emu)


@pytest.mark.gpu
def test_photometric_kernels_match_hand_derived_integers(gpu):
    _photometric_on_device(gpu)


def _tf_same_on_device(dev):
    """Product path: engine.conv_block / dwconv_block with the TF-SAME leading pad + given output size (efficientnet_lite3._conv_same)."""
    from riders_amd import engine
    from riders_amd.midas.efficientnet_lite3 import same_pad
    engine.set_compute_dtype("fp32")
    for key, hw, k in (("ones_4x4_k3_s2", 4, 3), ("ones_5x5_k3_s2", 5, 3), ("ones_4x4_k5_s2", 4, 5)):
        want = KAT["tf_same"][key]
        x = torch.ones((1, hw, hw, 8), dtype=torch.float32, device=dev)
        w = torch.zeros((8, 8, k, k), dtype=torch.float32, device=dev)
        w[0, 0] = 1.0                                            # output channel 0 = sum of input channel 0 over the window
        lead, o = same_pad(hw, k, 2)
        y = engine.conv_block(x, torch.nn.Parameter(w), stride=2, pad=lead, out_hw=(o, o))
        assert y[0, :, :, 0].cpu().int().tolist() == want, (key, "dense")
        wd = torch.ones((8, 1, k, k), dtype=torch.float32, device=dev)
        yd = engine.dwconv_block(x, torch.nn.Parameter(wd), stride=2, pad=lead, out_hw=(o, o))
        assert yd[0, :, :, 3].cpu().int().tolist() == want, (key, "depthwise")


def test_tf_same_convolutions_emu(emu):
    _tf_same_on_device(emu)


@pytest.mark.gpu
def test_tf_same_convolutions(gpu):
    _tf_same_on_device(gpu)
