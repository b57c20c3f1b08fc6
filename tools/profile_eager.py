"""Where does the HOST time of the unchanged-caller path go?  cProfile over eager RC-Net steps (torch.autograd + torch.optim.Adam + loss.item())."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riders_amd import engine, rcnet_main  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    engine.set_compute_dtype("bf16")
    cfg = rcnet_main.ZJU_CONFIG
    torch.manual_seed(0)
    model = rcnet_main.build_model(dev, cfg); model.train()
    batch = rcnet_main.synthetic_batch(8, 256, 512, cfg, device=dev)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4)

    def step():
        loss = rcnet_main.forward_loss(model, batch, cfg)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.item()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    print("eager unchanged-caller step: %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
