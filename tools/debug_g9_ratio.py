import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tests import parity_cases as P
from tests import parity_cases_sml as S
import tests.parity_cases as PC
rat = []
orig = PC.close
def close(a, b, tol=PC.TOL, what=""):
    a2 = a.detach().float().cpu().numpy().astype(np.float64) if torch.is_tensor(a) else np.asarray(a, np.float64)
    b2 = b.detach().float().cpu().numpy().astype(np.float64) if torch.is_tensor(b) else np.asarray(b, np.float64)
    err = np.abs(a2 - b2).max() / max(np.abs(b2).max(), 1e-6)
    rat.append((err / tol, err, tol, what))
S.close = close
try:
    S.sml_net_case("cuda:0")
except AssertionError as e:
    print("ASSERT", str(e)[:200])
rat.sort(reverse=True)
for r in rat[:8]:
    print("%.2f err %.2e tol %.2e %s" % r)
