"""bf16 SML: (a) HIP bf16 vs HIP fp32 (what bf16 storage costs by itself), (b) HIP bf16 vs the oracle with the same rounding points."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from riders_amd import engine
from riders_amd.midas.midas_net_custom import MidasNet_small_videpth
from tests import parity_cases_sml as S
from tests.golden.fill import fill_state_dict, rand_array
dev = torch.device("cuda:0")
for (B, H, W) in ((4, 128, 192), (16, 256, 512)):
    res = {}
    for mode in ("fp32", "bf16"):
        engine.set_compute_dtype(mode); engine.clear_caches()
        m = MidasNet_small_videpth(device=dev, min_pred=0.1, max_pred=255.0, in_channels=3)
        fill_state_dict(m, "g9.sml")
        x = torch.from_numpy(rand_array("s16.x", (B, 3, H, W), 1.0)).to(dev).requires_grad_()
        d = torch.from_numpy(rand_array("s16.d", (B, 1, H, W), 0.3, lo=0.05) + np.float32(0.02)).to(dev)
        m.train(); pred = m.forward(x, d)
        (pred * torch.from_numpy(rand_array("s16.w", (B, 1, H, W), 1.0)).to(dev)).sum().backward()
        m.eval()
        with torch.no_grad():
            pe = m.forward(x.detach(), d)
        res[mode] = (pred.detach().double().cpu(), x.grad.double().cpu(), pe.double().cpu())
    engine.set_compute_dtype("fp32")
    l2 = lambda a, b: float((a - b).norm() / b.norm())
    print("HIP bf16 vs HIP fp32 (B=%d %dx%d): pred L2 %.3e  dx L2 %.3e  eval pred L2 %.3e" % (B, H, W, l2(res["bf16"][0], res["fp32"][0]),
          l2(res["bf16"][1], res["fp32"][1]), l2(res["bf16"][2], res["fp32"][2])), flush=True)
for args in ((2, 64, 96), (4, 128, 192)):
    try:
        S.sml_net_bf16_case(dev, *args)
    except AssertionError as e:
        print("ASSERT", args, str(e)[:300], flush=True)
