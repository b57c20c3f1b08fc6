import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tests import parity_cases_sml as S
from riders_amd import engine, sml_main
from riders_amd.midas.midas_net_custom import MidasNet_small_videpth
from riders_amd.optim import FlatAdam
from tests.parity_cases_sml import fill_state_dict
dev=torch.device('cuda:0')
batch_cpu = sml_main.synthetic_batch(4, 128, 192, seed=41)
for mode in ("fp32","bf16","fp32b"):
    engine.set_compute_dtype(mode[:4]); engine.clear_caches()
    torch.manual_seed(0)
    m = MidasNet_small_videpth(device=dev, min_pred=0.1, max_pred=255.0, in_channels=3)
    fill_state_dict(m, "g9.sml"); m.train()
    opt = FlatAdam(m.parameters(), lr=2e-4)
    orr = sml_main.make_outlier_removal()
    batch = tuple(b.to(dev) for b in batch_cpu)
    if mode=="fp32b":  # perturb: different but equivalent order -> measure fp32 run-to-run chaos: tiny weight perturbation
        with torch.no_grad(): opt.flat_param.mul_(1.0+1e-6)
    c=[float(sml_main.train_step(m, opt, batch, outlier=orr)) for _ in range(200)]
    print(mode, " ".join("%.4f"%v for v in c[::10]))
