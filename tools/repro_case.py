"""Stress the bf16 exact conv cases (fault / flakiness hunting): repeats each case and counts mismatches per output."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import parity_cases as P
dev = torch.device("cuda:0")
cases = [dict(cin=64, cout=8, k=3, s=1, H=8, W=5, N=2), dict(cin=16, cout=16, k=3, s=1, H=17, W=9, N=1),
         dict(cin=16, cout=16, k=3, s=1, N=1, up=((4, 3), (9, 17)), cin2=16), dict(cin=64, cout=64, k=3, s=1, H=9, W=19, N=1),
         dict(cin=16, cout=1, k=3, s=1, H=130, W=100, N=24)]
junk = []
with P.force_patch_conv():
    for c in cases:
        bad = [0, 0, 0]
        for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
            junk.append(torch.full((1 + rep * 37 % 5000,), float("nan"), device=dev))   # perturb the allocator, poison freed memory
            if len(junk) > 8:
                junk.pop(0)
            r = P.bf16_exact_conv_case(dev, report=True, **c)
            for i in range(3):
                bad[i] += 0 if r[i] else 1
        print("case", c, "mismatches fwd/dgrad/wgrad:", bad, flush=True)
