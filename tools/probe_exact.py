import sys; sys.path.insert(0, '/root/repo')
import torch
from tests import parity_cases as P
dev = torch.device("cuda:0")
for kw in (dict(cin=256, cout=256, k=3, s=1, H=15, W=6, N=6, cin2=128), dict(cin=256, cout=256, k=3, s=1, N=6, up=((7, 3), (15, 6))),
           dict(cin=128, cout=128, k=3, s=1, H=30, W=12, N=6, cin2=128), dict(cin=384, cout=256, k=3, s=1, H=15, W=6, N=24)):
    try:
        P.bf16_exact_conv_case(dev, **kw); print("exact ok", kw)
    except AssertionError as e:
        print("FAIL", kw, str(e)[:200])
