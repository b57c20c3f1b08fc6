"""Where does the bf16 mode's gradient error come from?  ("localise first".)

Runs ONE RC-Net training forward + backward at configs[1] (B = 8, 256x512, R = 240) twice on the HIP path -- fp32 activations and the
driver-timed bf16 mode -- from the same seed, with engine.taps_enable(): every tapped tensor (encoder outputs, pooled maps, point-MLP output,
the token matrix after each transformer layer, the latent, the decoder stages) and the gradient the backward has accumulated for it are
compared stage by stage (relative L2, cosine), followed by the per-module parameter-gradient vectors the parity tests bound.

    python tools/grad_localise.py [--batch 8] [--opts name=value,...] [--out gpurun_out/grad_localise.txt]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from riders_amd import engine, rcnet_main  # noqa: E402


def module_grads(model):
    enc, dec = model.encoder, model.decoder
    groups = {"encoder_image": enc.encoder_image, "attention": enc.attention, "encoder_depth": enc.encoder_depth, "decoder": dec}
    return {k: torch.cat([p.grad.detach().float().reshape(-1) for p in m.parameters() if p.grad is not None]).cpu() for k, m in groups.items()}


def run(mode, batch, cfg, dev, opts, round_operands=False):
    """round_operands (with mode fp32): every weight and the image rounded to bf16 ONCE, everything computed and stored in fp32 -- what a
    perturbation of the size of one bf16 rounding of the operands does to the gradients of the fp32 network itself"""
    engine.set_compute_dtype(mode)
    engine.clear_caches()
    engine.apply_opts(opts)
    engine.taps_enable(True)
    try:
        torch.manual_seed(0)
        model = rcnet_main.build_model(dev, cfg)
        model.train()
        if round_operands:
            with torch.no_grad():
                for p_ in model.parameters():
                    p_.copy_(p_.to(torch.bfloat16).to(torch.float32))
            batch = (batch[0].to(torch.bfloat16).to(torch.float32),) + tuple(batch[1:])
        image, pts, rois, gt = rcnet_main.prepare_batch(batch)
        label, valid = engine.rcnet_labels(gt, pts, 0.5)
        logits = model.forward(image, pts, rois)
        loss, _ = model.compute_loss(logits, label, valid, 2.5)
        loss.backward()
        torch.cuda.synchronize()
        fwd, grad = engine.taps()
        return dict(fwd=dict(fwd), grad=dict(grad), logits=logits.detach().float().cpu(), loss=float(loss), pgrads=module_grads(model))
    finally:
        engine.taps_enable(False)
        engine.set_compute_dtype("fp32")
        engine.clear_caches()


def rel(a, b):
    a, b = a.reshape(-1).double(), b.reshape(-1).double()
    n = float(b.norm())
    return float((a - b).norm() / max(n, 1e-30)), float(torch.dot(a, b) / max(float(a.norm()) * n, 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--opts", default=os.environ.get("RIDERS_OPTS", ""))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "grad_localise.txt"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = rcnet_main.ZJU_CONFIG
    batch = rcnet_main.synthetic_batch(a.batch, 256, 512, cfg, seed=1234, device=dev)
    r32 = run("fp32", batch, cfg, dev, "")
    r16 = run("bf16", batch, cfg, dev, a.opts)
    rop = run("fp32", batch, cfg, dev, "", round_operands=True)
    lines = ["bf16 (opts: %s) against fp32 on the HIP path, B = %d; relative L2 / cosine.  Last two columns: the fp32 path with its weights and image rounded to bf16 ONCE "
             "(everything else fp32) against the plain fp32 path -- the network's own sensitivity to a perturbation of that size" % (a.opts or "default", a.batch),
             "loss %.6f / %.6f / %.6f (bf16 / fp32 / fp32 with rounded operands)" % (r16["loss"], r32["loss"], rop["loss"]),
             "logits                      %.3e  %.6f      | %.3e  %.6f" % (rel(r16["logits"], r32["logits"]) + rel(rop["logits"], r32["logits"])), "",
             "%-22s %-22s %-22s | %-22s %s" % ("tap", "forward", "gradient", "forward (operands)", "gradient (operands)")]
    order = ["enc.skip0", "enc.skip1", "enc.skip2", "enc.skip3", "enc.latent_image", "enc.skip0_pooled", "enc.skip1_pooled", "enc.skip2_pooled", "enc.skip3_pooled",
             "enc.latent_pooled", "enc.mlp_out", "tf.in"] + ["tf.layer%d" % i for i in range(8)] + ["enc.latent", "dec.deconv4", "dec.deconv3", "dec.deconv2", "dec.deconv1"]
    for k in order:
        f = "%.3e %.6f" % rel(r16["fwd"][k], r32["fwd"][k]) if k in r16["fwd"] and k in r32["fwd"] else "-"
        g = "%.3e %.6f" % rel(r16["grad"][k], r32["grad"][k]) if k in r16["grad"] and k in r32["grad"] else "-"
        f2 = "%.3e %.6f" % rel(rop["fwd"][k], r32["fwd"][k]) if k in rop["fwd"] and k in r32["fwd"] else "-"
        g2 = "%.3e %.6f" % rel(rop["grad"][k], r32["grad"][k]) if k in rop["grad"] and k in r32["grad"] else "-"
        lines.append("%-22s %-22s %-22s | %-22s %s" % (k, f, g, f2, g2))
    lines.append("")
    for k in r32["pgrads"]:
        lines.append("parameter gradient %-14s %.3e  %.6f      | %.3e  %.6f" % ((k,) + rel(r16["pgrads"][k], r32["pgrads"][k]) + rel(rop["pgrads"][k], r32["pgrads"][k])))
    txt = "\n".join(lines)
    print(txt)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        f.write(txt + "\n")


if __name__ == "__main__":
    main()
