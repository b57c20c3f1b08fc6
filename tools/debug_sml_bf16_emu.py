"""Block-by-block: HIP bf16 vs the oracle with the same rounding points (train and eval mode)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from riders_amd import engine
from riders_amd.midas.midas_net_custom import MidasNet_small_videpth
from riders_amd.midas import efficientnet_lite3 as E
from oracle import sml as OS, effnet_lite3_torch as OE
from tests.golden.fill import fill_state_dict, rand_array
dev = torch.device("cuda:0")
B, H, W = 4, 128, 192
xin = rand_array("s16.x", (B, 3, H, W), 1.0); din = rand_array("s16.d", (B, 1, H, W), 0.3, lo=0.05) + np.float32(0.02)
engine.set_compute_dtype("bf16"); OE.Precision.bf16 = True
m = MidasNet_small_videpth(device=dev, min_pred=0.1, max_pred=255.0, in_channels=3)
sd = fill_state_dict(m, "g9.sml")
o = OS.SMLOracle(); o.load_state_dict({k: v.cpu() for k, v in sd.items()})
for mode in ("eval", "train"):
    getattr(m, mode)(); getattr(o, mode)()
    hip, ref = [], []
    blocks_h = [b for n, b in m.named_modules() if isinstance(b, (E.InvertedResidual, E.DepthwiseSeparableConv))]
    blocks_o = [b for n, b in o.named_modules() if isinstance(b, (OE.InvertedResidual, OE.DepthwiseSeparableConv))]
    names = [n for n, b in o.named_modules() if isinstance(b, (OE.InvertedResidual, OE.DepthwiseSeparableConv))]
    origs = []
    for b in blocks_h:
        f = b._fwd
        origs.append((b, f))
        b._fwd = (lambda f: (lambda x: (lambda out: (hip.append(out.float().permute(0, 3, 1, 2).cpu().clone()), out)[1])(f(x))))(f)
    hooks = [b.register_forward_hook(lambda mod, i, out: ref.append(out.detach().clone())) for b in blocks_o]
    with torch.no_grad():
        ph = m.forward(torch.from_numpy(xin).to(dev), torch.from_numpy(din).to(dev))
        po = o(torch.from_numpy(xin), torch.from_numpy(din))
    for b, f in origs:
        del b._fwd
    for h in hooks:
        h.remove()
    print("----", mode)
    for n, a, b in zip(names, hip, ref):
        print("%-28s %-18s relL2 %.3e" % (n, tuple(b.shape), float((a - b).norm() / b.norm())))
    print("pred relL2 %.3e" % float((ph.float().cpu() - po).norm() / po.norm()))
