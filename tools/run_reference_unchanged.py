"""BUILD CONTAINER ONLY -- executes the claim of INTEGRATION.md section 1: the reference's OWN training entry point
(/root/reference/RCNet/rcnet_main.py, `train(...)`, unmodified, imported where it lies -- nothing is copied) runs on the riders_amd modules
when they are aliased under the reference's module names, and produces the losses the reference's own modules produce.

    python tools/run_reference_unchanged.py            # runs both variants as child processes and compares them

Both variants run the SAME script on the SAME files with the SAME seeds and restore the SAME initial checkpoint (written by the reference's
RCNetModel.save_model, read back through `restore_path` = checkpoint interchange):
  ref : the reference's networks / linear_attention / rcnet_model / utils.net_utils (PyTorch CPU);
  hip : the aliasing block of INTEGRATION.md section 1, the kernels being the host fiber-emulator build of riders_amd/csrc (tests/emu; there is
        no GPU here).  On a GPU box the same aliasing binds libriders_hip.so.
torchvision / cv2 / tensorboard are absent in this image: tests/golden/stubs supplies import stubs (as for the golden generators).
Never run on the GPU box (/root/reference does not exist there) and not part of the test suite; the log is committed under profiles/.
"""
import json
import os
import subprocess
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH, K, H, W = [64, 32], 2, 64, 96


def write_dataset(root):
    """One synthetic frame in the reference's on-disk layout: <root>/scene0/{image,radar,gt}/000.* (RGB PNG, (N,3) radar .npy, 16-bit depth PNG)."""
    import numpy as np
    from PIL import Image
    from data import data_utils as RU      # the reference's own writer
    rs = np.random.RandomState(4242)
    for sub in ("image", "radar", "gt"):
        os.makedirs(os.path.join(root, "scene0", sub))
    img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    gt = np.zeros((H, W), np.float32)
    idx = rs.choice(H * W, 2500, replace=False)
    gt.flat[idx] = rs.uniform(0.5, 70.0, idx.size).astype(np.float32)
    radar = np.stack([rs.uniform(0, W - 1, 9), rs.uniform(0, H - 1, 9), rs.uniform(2, 60, 9)], 1)
    Image.fromarray(img).save(os.path.join(root, "scene0", "image", "000.png"))
    np.save(os.path.join(root, "scene0", "radar", "000.npy"), radar)
    RU.save_depth(gt, os.path.join(root, "scene0", "gt", "000.png"))


def child(mode, workdir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden", "stubs"), REF, os.path.join(REF, "RCNet")]
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter(object):
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None
    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    import random
    import numpy as np
    import torch
    torch.set_num_threads(8)
    if mode == "hip":
        # ---- INTEGRATION.md section 1, verbatim ----------------------------------------------------------------------------------------
        import riders_amd.net_utils, riders_amd.networks, riders_amd.linear_attention, riders_amd.rcnet_model
        sys.modules['networks'] = riders_amd.networks                 # RCNet/rcnet_model.py:3
        sys.modules['linear_attention'] = riders_amd.linear_attention  # RCNet/networks.py:4
        sys.modules['rcnet_model'] = riders_amd.rcnet_model            # RCNet/rcnet_main.py:7
        import utils; utils.net_utils = riders_amd.net_utils           # RCNet/networks.py:2
        # ---- the kernels: no GPU in the build container -> the emulator build of the same sources (tests only) ---------------------------
        from riders_amd import _lib, engine
        from tests.emu import build_emu
        _lib._install_for_tests(build_emu.build())
        engine.set_compute_dtype("fp32")
    import rcnet_main      # the reference's script, unmodified
    assert rcnet_main.__file__.startswith(REF), rcnet_main.__file__
    assert (rcnet_main.RCNetModel.__module__ == "riders_amd.rcnet_model") == (mode == "hip"), rcnet_main.RCNetModel.__module__
    data_root = os.path.join(workdir, "data")
    settings = dict(
        root=data_root, scenes=["scene0"], image_file="image", radar_file="radar", gt_file="gt",
        batch_size=1, patch_size=PATCH, total_points_sampled=K, sample_probability_of_lidar=0.0, normalized_image_range=[0, 1],
        encoder_type=['rcnet', 'batch_norm'], n_filters_encoder_image=[32, 64, 128, 128, 128], n_neurons_encoder_depth=[32, 64, 128, 128, 128],
        decoder_type=['multiscale', 'batch_norm'], n_filters_decoder=[256, 128, 64, 32, 16],
        weight_initializer='kaiming_uniform', activation_func='leaky_relu',
        learning_rates=[2e-4], learning_schedule=[2], augmentation_probabilities=[0.0], augmentation_schedule=[-1],
        augmentation_random_brightness=[-1, -1], augmentation_random_contrast=[-1, -1], augmentation_random_saturation=[-1, -1],
        augmentation_random_noise_type=['none'], augmentation_random_noise_spread=-1, augmentation_random_flip_type=['none'],
        w_weight_decay=0.0, w_positive_class=2.5, max_distance_correspondence=0.5, set_invalid_to_negative_class=False,
        checkpoint_dirpath=os.path.join(workdir, "ckpt_" + mode), n_step_per_summary=10 ** 9, n_step_per_checkpoint=10 ** 9,
        restore_path=os.path.join(workdir, "init.pth"), n_thread=0)
    if mode == "init":      # the shared starting point, written by the REFERENCE's classes
        torch.manual_seed(7)
        m = rcnet_main.RCNetModel(3, 3, PATCH, settings["encoder_type"], settings["n_filters_encoder_image"], settings["n_neurons_encoder_depth"],
                                  settings["decoder_type"], settings["n_filters_decoder"], device=torch.device("cpu"))
        m.data_parallel()      # train() wraps before it restores: its checkpoints carry the `module.` prefix (rcnet_model.py:259-265)
        opt = torch.optim.Adam([{'params': m.parameters(), 'weight_decay': 0.0}], lr=2e-4)
        m.save_model(settings["restore_path"], 0, opt)
        write_dataset(data_root)
        return
    losses = []
    real_print = print

    def spy(*a, **k):      # the script prints "<step>/<steps> epoch:<e>: <loss>" per step (rcnet_main.py:352)
        s = " ".join(str(x) for x in a)
        if " epoch:" in s:
            losses.append(float(s.rsplit(":", 1)[1]))
        real_print(*a, **k)
    rcnet_main.print = spy
    torch.manual_seed(11); np.random.seed(11); random.seed(11)
    rcnet_main.train(**settings)
    out = dict(mode=mode, losses=losses, checkpoint=sorted(os.listdir(settings["checkpoint_dirpath"])))
    ck = torch.load(os.path.join(settings["checkpoint_dirpath"], "model-2.pth"), map_location="cpu")
    out["train_step"] = int(ck["train_step"])
    out["encoder_keys"] = len(ck["radarnet_encoder_state_dict"])
    # the two optimizer steps, as one vector: final minus initial weights (Adam normalises every element's update to ~lr, so elements whose
    # gradient is numerically zero may differ in sign between two fp32 implementations; the comparison is a relative L2 of the whole update)
    init = torch.load(settings["restore_path"], map_location="cpu")
    upd = torch.cat([(ck[k][n].double() - init[k][n].double()).reshape(-1) for k in ("radarnet_encoder_state_dict", "radarnet_decoder_state_dict")
                     for n in ck[k] if ck[k][n].is_floating_point() and "running" not in n])
    torch.save(upd, os.path.join(workdir, mode + "_update.pt"))
    json.dump(out, open(os.path.join(workdir, mode + ".json"), "w"))


def main():
    if len(sys.argv) == 3:
        return child(sys.argv[1], sys.argv[2])
    if not os.path.isdir(REF):
        sys.exit("run_reference_unchanged.py needs /root/reference (build container only)")
    with tempfile.TemporaryDirectory() as wd:
        for mode in ("init", "ref", "hip"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), mode, wd], capture_output=True, text=True)
            tail = "\n".join(l for l in r.stdout.splitlines() if " epoch:" in l or "Begin training" in l)
            print("[%s] rc=%d\n%s" % (mode, r.returncode, tail), flush=True)
            if r.returncode != 0:
                sys.exit(r.stderr[-3000:])
        ref, hip = (json.load(open(os.path.join(wd, m + ".json"))) for m in ("ref", "hip"))
        import torch
        u_ref, u_hip = (torch.load(os.path.join(wd, m + "_update.pt")) for m in ("ref", "hip"))
        upd_err = float((u_ref - u_hip).norm() / u_ref.norm())
        upd_cos = float(torch.dot(u_ref, u_hip) / (u_ref.norm() * u_hip.norm()))
    print("reference modules :", ref)
    print("riders_amd modules:", hip)
    assert len(ref["losses"]) == len(hip["losses"]) == 2 and ref["train_step"] == hip["train_step"] == 2
    assert ref["checkpoint"] == hip["checkpoint"] and ref["encoder_keys"] == hip["encoder_keys"]
    for a, b in zip(ref["losses"], hip["losses"]):
        assert abs(a - b) <= 1e-3 * abs(a), (a, b)
    print("two-step weight update (final - initial, %d elements): relative L2 difference %.3e, cosine %.6f" % (u_ref.numel(), upd_err, upd_cos))
    assert upd_cos > 0.99, upd_cos
    print("OK: the reference's rcnet_main.train() ran unchanged on the aliased riders_amd modules; per-step losses within 1e-3 of the reference's "
          "own modules, the saved checkpoints have the same files / keys / step count")


if __name__ == "__main__":
    main()
