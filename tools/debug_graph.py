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
step = rcnet_main.GraphedTrainStep(model, opt, batch, cfg)
for i in range(14):
    loss = step()
    torch.cuda.synchronize()
    gfin = bool(torch.isfinite(opt.flat_grad).all()); pfin = bool(torch.isfinite(opt.flat_param).all())
    print("replay", i, "loss", float(loss), "grads finite", gfin, "params finite", pfin, flush=True)
