import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riders_amd import engine, rcnet_main
from riders_amd.optim import FlatAdam
dev = torch.device("cuda:0")
engine.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "bf16")
cfg = rcnet_main.ZJU_CONFIG
torch.manual_seed(0)
model = rcnet_main.build_model(dev, cfg); model.train()
opt = FlatAdam(model.parameters(), lr=cfg['learning_rate'])
batch = rcnet_main.synthetic_batch(8, 256, 512, cfg, seed=1234, device=dev)
names = {id(p): n for net, pre in ((model.encoder, "enc."), (model.decoder, "dec.")) for n, p in net.named_parameters(prefix=pre[:-1])}
for step in range(16):
    loss = rcnet_main.compute_gradients(model, opt, batch, cfg)
    bad = [names[id(p)] for p in model.parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    gn = sum(float(p.grad.double().pow(2).sum()) for p in model.parameters() if p.grad is not None) ** 0.5
    print("step", step, "loss", float(loss), "gradnorm", gn, "nonfinite grads:", bad[:6], flush=True)
    if bad: break
    opt.step()
    badp = [names[id(p)] for p in model.parameters() if not torch.isfinite(p).all()]
    if badp:
        print("nonfinite params", badp[:6]); break
