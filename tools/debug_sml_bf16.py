"""Layer-by-layer bf16 vs fp32 comparison of the SML forward (same weights, same input): where does the bf16 error jump?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from riders_amd import engine
from riders_amd.midas.midas_net_custom import MidasNet_small_videpth
from riders_amd.midas import efficientnet_lite3 as E
from tests.golden.fill import fill_state_dict, rand_array
dev = torch.device("cuda:0")
B, H, W = 2, 64, 96
x = torch.from_numpy(rand_array("g9.x", (B, 3, H, W), 1.0)).to(dev)
d = torch.from_numpy(rand_array("g9.d", (B, 1, H, W), 0.3, lo=0.05) + np.float32(0.02)).to(dev)
acts = {}
orig_conv, orig_dw, orig_bil, orig_act = engine.conv_block, engine.dwconv_block, engine.bilinear2x, engine.activation
def run(mode):
    engine.set_compute_dtype(mode)
    engine.clear_caches()
    m = MidasNet_small_videpth(device=dev, min_pred=0.1, max_pred=255.0, in_channels=3)
    fill_state_dict(m, "g9.sml")
    m.train()
    rec = []
    def wrap(name, fn):
        def f(*a, **k):
            out = fn(*a, **k)
            rec.append((name + " " + "x".join(map(str, out.shape)) + (" k%d" % a[1].shape[-1] if len(a) > 1 and torch.is_tensor(a[1]) and a[1].dim() == 4 else "")
                        + (" bn" if k.get("bn") is not None else "") + (" bias" if k.get("bias") is not None else "") + (" res" if k.get("residual") is not None else "")
                        + (" s%d" % k["stride"] if "stride" in k else ""), out.float().clone()))
            return out
        return f
    engine.conv_block = wrap("conv", orig_conv); engine.dwconv_block = wrap("dw", orig_dw)
    engine.bilinear2x = wrap("bilinear", orig_bil); engine.activation = wrap("act", orig_act)
    E.engine = engine
    with torch.no_grad():
        pred = m.forward(x, d)
    rec.append(("pred", pred.float().clone()))
    engine.conv_block, engine.dwconv_block, engine.bilinear2x, engine.activation = orig_conv, orig_dw, orig_bil, orig_act
    return rec
r32 = run("fp32"); r16 = run("bf16")
for (n, a), (_, b) in zip(r32, r16):
    err = float((a - b).norm() / a.norm().clamp_min(1e-20)); mx = float((a - b).abs().max() / a.abs().max().clamp_min(1e-20))
    print("%-50s relL2 %.3e  max %.3e  |ref|max %.3e" % (n, err, mx, float(a.abs().max())))
