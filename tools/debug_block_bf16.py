"""Inside one InvertedResidual block: HIP bf16 vs rounding-emulating oracle after every sub-step (eval mode, so no batch statistics)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from riders_amd import engine
from riders_amd.engine import ACT_NONE, ACT_RELU6
from riders_amd.midas import efficientnet_lite3 as E
from oracle import effnet_lite3_torch as OE
from tests.golden.fill import fill_state_dict, rand_array
dev = torch.device("cuda:0")
engine.set_compute_dtype("bf16"); OE.Precision.bf16 = True
R = OE.Precision.r
for (cin, cout, k, s, H, W, training) in ((24, 32, 3, 2, 64, 96, False), (24, 32, 3, 2, 64, 96, True), (32, 32, 3, 1, 32, 48, False)):
    mine = E.InvertedResidual(cin, cout, k, s).to(dev); ref = OE.InvertedResidual(cin, cout, k, s)
    ref.load_state_dict({kk: v.cpu() for kk, v in fill_state_dict(mine, "dbg").items()})
    mine.train(training); ref.train(training)
    x = torch.from_numpy(rand_array("dbg.x", (4, cin, H, W), 3.0, lo=0.0)).bfloat16().float()
    xh = x.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16()
    def l2(a, b): return float((a.float().permute(0, 3, 1, 2).cpu() - b).norm() / b.norm())
    with torch.no_grad():
        # step 1: pw conv only (no bn): y
        y1h = engine.conv_block(xh, mine.conv_pw.weight, stride=1, pad=0)
        y1o = ref.conv_pw(x)
        print("cfg", (cin, cout, k, s, training), "pw conv y           %.3e" % l2(y1h, y1o))
        h1h = E._conv_same(xh, mine.conv_pw, mine.bn1, ACT_RELU6, training)
        h1o = ref.act1(ref.bn1(ref.conv_pw(x)))
        print("   pw+bn+relu6          %.3e" % l2(h1h, h1o))
        # feed the ORACLE's h1 to both from here on, so that errors do not accumulate
        h1in = h1o.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16()
        y2h = engine.dwconv_block(h1in, mine.conv_dw.weight, stride=s, pad=E.same_pad(H, k, s)[0], out_hw=(E.same_pad(H, k, s)[1], E.same_pad(W, k, s)[1]))
        y2o = ref.conv_dw(h1o)
        print("   dw conv y            %.3e" % l2(y2h, y2o))
        h2h = E._dw_same(h1in, mine.conv_dw, mine.bn2, ACT_RELU6, training)
        h2o = ref.act2(ref.bn2(ref.conv_dw(h1o)))
        print("   dw+bn+relu6          %.3e" % l2(h2h, h2o))
        h2in = h2o.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16()
        y3h = engine.conv_block(h2in, mine.conv_pwl.weight, stride=1, pad=0)
        y3o = ref.conv_pwl(h2o)
        print("   pwl conv y           %.3e   (K=%d)" % (l2(y3h, y3o), h2o.shape[1]))
        h3h = E._conv_same(h2in, mine.conv_pwl, mine.bn3, ACT_NONE, training)
        h3o = R(ref.bn3(ref.conv_pwl(h2o)))
        print("   pwl+bn               %.3e" % l2(h3h, h3o))
        d = (y3h.float().permute(0, 3, 1, 2).cpu() - y3o)
        nz = (d != 0).float().mean()
        print("   pwl y: fraction of elements differing %.4f, max |diff|/|ref| %.3e" % (float(nz), float((d.abs() / y3o.abs().clamp_min(1e-6)).max())))
