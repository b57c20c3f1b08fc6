"""A/B of the batched weight re-pack (rd_conv_pack_weights_batch): element-wise form against the 16-byte-unit form, on the cached operands of
the RC-Net and the SML after one training step each (bf16).   python tools/bench_pack.py"""
import contextlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riders_amd import engine, rcnet_main, sml_main  # noqa: E402
from riders_amd.optim import FlatAdam  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    engine.set_compute_dtype("bf16")
    for name in ("rcnet", "sml"):
        engine.clear_caches()
        torch.manual_seed(0)
        if name == "rcnet":
            cfg = rcnet_main.ZJU_CONFIG
            m = rcnet_main.build_model(dev, cfg); m.train()
            opt = FlatAdam(m.parameters(), lr=1e-4)
            batch = rcnet_main.synthetic_batch(2, 128, 256, cfg, device=dev)
            rcnet_main.train_step(m, opt, batch, cfg)
        else:
            with contextlib.redirect_stdout(sys.stderr):
                m = sml_main.build_model(dev); m.train()
            opt = FlatAdam(m.parameters(), lr=1e-4)
            batch = sml_main.synthetic_batch(2, 128, 256, device=dev)
            sml_main.train_step(m, opt, batch, outlier=sml_main.make_outlier_removal())
        torch.cuda.synchronize()
        for rep in range(2):
            for vec, bmap in ((0, 0), (1, 0), (0, 1), (1, 1)):
                engine.set_switch("pack_vec", vec); engine.set_switch("pack_map", bmap)
                engine.refresh_packed(opt._owner)      # builds the device tables of this variant
                tab = engine._pack_table[opt._owner]
                lib, st = engine.L(), engine._stream(opt.flat_param)

                def launch():      # the launch alone (refresh_packed's host-side walk over the cache costs ~100 us of Python per call)
                    if bmap:
                        lib.rd_conv_pack_weights_batch_map(engine._p(tab["dev"]), tab["n"], engine.RD_BF16, engine._p(tab["map"]), tab["blocks"], st)
                    else:
                        lib.rd_conv_pack_weights_batch_half(engine._p(tab["dev"]), tab["n"], engine.RD_BF16, st)
                for _ in range(100):
                    launch()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(400):
                    launch()
                e.record()
                torch.cuda.synchronize()
                print("%s pack_vec=%d pack_map=%d  %.1f us per launch (%d items, %d blocks)" % (name, vec, bmap, s.elapsed_time(e) * 1e3 / 400, tab["n"], tab["blocks"] if bmap else 256 * tab["n"]), flush=True)
        engine.set_switch("pack_vec", 1); engine.set_switch("pack_map", 1)


if __name__ == "__main__":
    main()
