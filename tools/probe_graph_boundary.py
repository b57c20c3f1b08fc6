"""How much of a graphed training step is the boundary between the hipGraph replay and the eager Adam / re-pack launches?
Times (a) the full step, (b) the graph replays alone, (c) Adam + re-pack alone (HIP events around 100 iterations each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riders_amd import engine, rcnet_main, sml_main
from riders_amd.optim import FlatAdam
dev = torch.device("cuda:0")
engine.set_compute_dtype("bf16")


def timed(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name in ("rcnet", "sml"):
    engine.clear_caches()
    torch.manual_seed(0)
    if name == "rcnet":
        cfg = rcnet_main.ZJU_CONFIG
        model = rcnet_main.build_model(dev, cfg); model.train()
        batch = rcnet_main.synthetic_batch(8, 256, 512, cfg, seed=1, device=dev)
        opt = FlatAdam(model.parameters(), lr=2e-4)
        step = rcnet_main.GraphedTrainStep(model, opt, batch, cfg)
    else:
        cfg = sml_main.ZJU_SML_CONFIG
        model = sml_main.build_model(dev, cfg); model.train()
        batch = sml_main.synthetic_batch(16, 256, 512, seed=1, device=dev)
        opt = FlatAdam(model.parameters(), lr=1e-4)
        step = sml_main.GraphedTrainStep(model, opt, batch, cfg, outlier=sml_main.make_outlier_removal(cfg))
    full = timed(step)

    def replay_only():
        for g in step.graphs:
            g.replay()

    def adam_only():
        step.opt.mark_touched(step.touched)
        step.opt.step()
    a, b, c = full, timed(replay_only), timed(adam_only)
    print("%s: full step %.3f ms | graph replays alone %.3f ms | Adam + re-pack alone %.3f ms | boundary = full - (replay + adam) = %+.3f ms" % (name, a, b, c, a - b - c), flush=True)
    del step, opt, model
    engine.set_param_grad_allocator(None)
