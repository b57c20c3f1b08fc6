"""bf16 layer-level cases against the rounding-emulating oracle: print the achieved errors (loose tolerances here; the tests pin them)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import parity_cases as P, parity_cases_sml as S
dev = torch.device("cuda:0")
import tests.parity_cases as PC
orig_close, orig_l2 = PC.close, PC.close_l2
def wrap(fn):
    def f(a, b, tol, what=""):
        aa = a.detach().float().cpu().double() if torch.is_tensor(a) else torch.from_numpy(np.asarray(a, np.float64))
        bb = b.detach().float().cpu().double() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b, np.float64))
        print("  %-46s max %.2e  L2 %.2e  |ref|max %.2e rms %.2e" % (what, float((aa - bb).abs().max() / bb.abs().max().clamp_min(1e-6)),
              float((aa - bb).norm() / bb.norm().clamp_min(1e-30)), float(bb.abs().max()), float(bb.pow(2).mean().sqrt())))
        return 0.0
    return f
PC.close = wrap(orig_close); PC.close_l2 = wrap(orig_l2); S.close = PC.close; S.close_l2 = PC.close_l2
which = sys.argv[1] if len(sys.argv) > 1 else "all"
with P.bf16_mode():
    if which in ("all", "dec"):
        print("decoder_block wide"); P.decoder_block_case(dev, cin=256, cskip=128, cout=256, hs=(7, 3), hv=(15, 6), N=6, tol=1e9)
        print("decoder_block wide, more samples"); P.decoder_block_case(dev, cin=256, cskip=128, cout=256, hs=(7, 3), hv=(15, 6), N=48, tol=1e9)
P.Precision = None
print("fp32 reference run of the same case")
P.decoder_block_case(dev, cin=256, cskip=128, cout=256, hs=(7, 3), hv=(15, 6), N=6, tol=1e9)
