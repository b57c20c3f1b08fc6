import sys; import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from riders_amd import _lib, engine
import os
DEV = "cuda:0"
from riders_amd.engine import _p, L
lib = L()
for (N, H, W, C, k, s, p) in [(2, 2, 3, 1392, 3, 1, 1), (2, 2, 3, 816, 5, 1, 2), (2, 4, 6, 144, 3, 2, 0), (2, 5, 7, 48, 5, 1, 2)]:
    OH, OW = (H + s - 1) // s, (W + s - 1) // s
    torch.manual_seed(0)
    x = (torch.randn(N, H, W, C) + 3.0).to(DEV)
    w = torch.randn(C, 1, k, k).to(DEV)
    y = torch.empty(N, OH, OW, C, device=DEV); y2 = torch.empty_like(y)
    rows = lib.rd_dwconv_stats_rows(N, OH, OW, C, k, s)
    st = torch.full((rows, C, 2), 7.0, device=DEV)
    assert lib.rd_dwconv_fwd_stats(_p(x), _p(w), _p(y), _p(st), N, H, W, C, OH, OW, k, s, p, 0, None) == 0
    assert lib.rd_dwconv_fwd(_p(x), _p(w), _p(y2), N, H, W, C, OH, OW, k, s, p, 0, None) == 0
    torch.cuda.synchronize(); y, y2, st = y.cpu(), y2.cpu(), st.cpu()
    print("y equal", torch.equal(y, y2), "rows", rows)
    ref1 = y.double().sum((0, 1, 2)); ref2 = (y.double() ** 2).sum((0, 1, 2))
    got = st.double().sum(0)
    print("  sum err", float((got[:, 0] - ref1).abs().max() / ref1.abs().max()), "sq err", float((got[:, 1] - ref2).abs().max() / ref2.abs().max()))
