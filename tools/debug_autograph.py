import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riders_amd import engine, rcnet_main
gpu = torch.device("cuda:0")
cfg = dict(rcnet_main.ZJU_CONFIG, patch_size=[64, 32], total_points_sampled=4)
batches = [rcnet_main.synthetic_batch(2, 64, 96, cfg, seed=20 + i, device=gpu) for i in range(5)]
res = {}
for mode in ("eager", "autograph", "manual"):
    engine.set_autograph(mode == "autograph")
    engine.set_deterministic_roi_pool(True)
    torch.manual_seed(0)
    model = rcnet_main.build_model(gpu, cfg); model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    rec = []
    for i, b in enumerate(batches):
        loss = rcnet_main.forward_loss(model, b, cfg)
        if mode == "manual" and i == 3:
            keep = [None if p.grad is None else p.grad.detach().clone() for p in model.parameters()]
            opt.zero_grad()
            loss.backward()
            for p, k in zip(model.parameters(), keep):
                if k is not None and p.grad is not None:
                    p.grad = p.grad + k
        else:
            if i != 3:
                opt.zero_grad()
            loss.backward()
        names = [n for n, _ in list(model.encoder.named_parameters()) + list(model.decoder.named_parameters())]
        grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in zip(names, model.parameters())}
        opt.step()
        rec.append((loss.item(), grads))
    res[mode] = rec
    engine.set_autograph(False)
for ref, other in (("manual", "eager"), ("manual", "autograph")):
  print(ref, "vs", other)
  for i in range(5):
    le, ge = res[ref][i]; la, ga = res[other][i]
    worst = max(((float((ge[k] - ga[k]).abs().max() / (ge[k].abs().max() + 1e-12)), k) for k in ge if ge[k] is not None and ga[k] is not None), default=(0, ""))
    mism = [k for k in ge if (ge[k] is None) != (ga[k] is None)]
    print(i, le, la, "worst rel grad diff", worst, "none-mismatch", mism[:4])
i = 3
ge = res["manual"][i][1]; ga = res["autograph"][i][1]
g2 = res["manual"][2][1]
for k in ("attention.layers.5.q_proj.weight", "encoder_depth.mlp.0.fully_connected.bias", "deconv4.conv.conv.weight", "output0.conv.weight"):
    a, b, c = ge[k], ga[k], g2[k]
    g3 = a - c
    print(k, "max|expected| %.3e max|autograph| %.3e  |auto-expected| %.3e  |auto-g3| %.3e |auto-(2g3)| %.3e |auto - (g2+g2)| %.3e  nan %s" % (
        float(a.abs().max()), float(b.abs().max()), float((b - a).abs().max()), float((b - g3).abs().max()), float((b - 2 * g3).abs().max()), float((b - 2 * c).abs().max()), bool(torch.isnan(b).any())))
bad = [k for k in ge if ge[k] is not None and float((ge[k] - ga[k]).abs().max()) > 1e-6 * float(ge[k].abs().max() + 1e-30)]
print(len(bad), "of", len([k for k in ge if ge[k] is not None]), "params differ:", bad[:60])
