"""BUILD CONTAINER ONLY -- the Scale-Map-Learner half of INTEGRATION.md section 1, executed: the reference's OWN entry points
`/root/reference/train_zju.py: train(...)` (two optimisation steps) and `/root/reference/val_zju.py: validate(...)` (one frame), unmodified and
imported where they lie (nothing is copied), run once on the reference's modules and once with riders_amd's modules aliased under the
reference's module names:

    modules.midas.midas_net_custom  -> riders_amd.midas.midas_net_custom     (train_zju.py:14, val_zju.py:17)
    utils.loss                      -> riders_amd.loss                       (train_zju.py:12)
    utils.net_utils (OutlierRemoval)-> riders_amd.net_utils                  (train_zju.py:10)

    python tools/run_reference_sml_unchanged.py        # runs init / ref / hip as child processes and compares them

Both variants read the SAME files (written in the reference's on-disk layout by the `init` child), use the SAME seeds and load the SAME initial
state_dict (written by the reference's own MidasNet_small_videpth.save) through `restore_path`.  In this container there is no GPU: the hip
variant's kernels are the host fiber-emulator build of riders_amd/csrc (tests/emu); on a GPU box the same aliasing binds libriders_hip.so.
Absent third-party packages are stubbed (tests/golden/stubs: torchvision, cv2 -- imread / cvtColor / INTER_NEAREST resize written here from
their documented behaviour --, tensorboard; `modules.midas.dpt_depth`, an import-time-only dependency on timm, is replaced by an empty module);
`torch.hub.load` (network) returns the oracle's restatement of tf_efficientnet_lite3 on the reference side (third-party, unpinned).
Never run on the GPU box, not part of the test suite; the log is committed under profiles/.
"""
import json
import os
import subprocess
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
H, W = 72, 96      # the reference's transform resizes every frame to a multiple of 32 near 288: 288 x 384 net input (transforms.py:317-326)


def write_dataset(root, result_root):
    """Two frames in the reference's layout: <root>/scene0/{thermal_undistort,any,radar_png,lidar_png,lidar_png_int}/00N.png + <result_root>/rcnet/scene0/depth_predicted."""
    import numpy as np
    from PIL import Image
    from data import data_utils as RU      # the reference's own 16-bit depth writer
    rs = np.random.RandomState(777)
    sc = os.path.join(root, "scene0")
    for sub in ("thermal_undistort", "any", "radar_png", "lidar_png", "lidar_png_int"):
        os.makedirs(os.path.join(sc, sub))
    rc = os.path.join(result_root, "rcnet", "scene0", "depth_predicted")
    os.makedirs(rc)
    for i in range(2):
        name = "%03d.png" % i
        Image.fromarray(rs.randint(0, 256, (H, W, 3)).astype(np.uint8)).save(os.path.join(sc, "thermal_undistort", name))
        depth = rs.uniform(2.0, 60.0, (H, W)).astype(np.float32)
        mono = (1.0 / depth) * rs.uniform(8.0, 12.0) * (1.0 + 0.05 * rs.randn(H, W))       # relative inverse depth, "any" (scaled to 16 bits)
        RU.save_depth(np.clip(mono, 0.01, 200).astype(np.float32), os.path.join(sc, "any", name))
        radar = np.where(rs.rand(H, W) < 0.02, depth * (1 + 0.05 * rs.randn(H, W)), 0).astype(np.float32)
        RU.save_depth(radar, os.path.join(sc, "radar_png", name))
        lidar = np.where(rs.rand(H, W) < 0.1, depth, 0).astype(np.float32)
        RU.save_depth(lidar, os.path.join(sc, "lidar_png", name))
        dense = np.where(rs.rand(H, W) < 0.8, depth * (1 + 0.02 * rs.randn(H, W)), 0).astype(np.float32)      # interpolated lidar: dense, some holes
        RU.save_depth(dense, os.path.join(sc, "lidar_png_int", name))
        rcn = np.where(rs.rand(H, W) < 0.25, depth * (1 + 0.1 * rs.randn(H, W)), 0).astype(np.float32)
        RU.save_depth(np.clip(rcn, 0, 200), os.path.join(rc, name))


def child(mode, workdir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden", "stubs"), REF]
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter(object):
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None
    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    dpt = types.ModuleType("modules.midas.dpt_depth")      # import-time-only dependency on timm (train_zju.py:15, val_zju.py:18); never constructed
    dpt.DPTDepthModel = type("DPTDepthModel", (), {})
    sys.modules["modules.midas.dpt_depth"] = dpt
    import random
    import numpy as np
    import torch
    torch.set_num_threads(8)
    from oracle.effnet_lite3_torch import EfficientNetLite3Features
    torch.hub.load = lambda *a, **k: EfficientNetLite3Features()      # the reference fetches geffnet from the network (modules/midas/blocks.py:45-50)
    if mode == "hip":
        # ---- INTEGRATION.md section 1 (SML block), verbatim -----------------------------------------------------------------------------
        import riders_amd.midas.midas_net_custom, riders_amd.loss, riders_amd.net_utils
        sys.modules['modules.midas.midas_net_custom'] = riders_amd.midas.midas_net_custom   # train_zju.py:14, val_zju.py:17
        sys.modules['utils.loss'] = riders_amd.loss                                          # train_zju.py:12
        sys.modules['utils.net_utils'] = riders_amd.net_utils                                # train_zju.py:10
        # ---- the kernels: no GPU in the build container -> the emulator build of the same sources (tests only) ---------------------------
        from riders_amd import _lib, engine
        from tests.emu import build_emu
        _lib._install_for_tests(build_emu.build())
        engine.set_compute_dtype("fp32")
    import train_zju      # the reference's scripts, unmodified
    import val_zju
    assert train_zju.__file__.startswith(REF) and val_zju.__file__.startswith(REF)
    assert (train_zju.MidasNet_small_videpth.__module__ == "riders_amd.midas.midas_net_custom") == (mode == "hip")
    assert (train_zju.compute_loss.__module__ == "riders_amd.loss") == (mode == "hip")
    assert (train_zju.OutlierRemoval.__module__ == "riders_amd.net_utils") == (mode == "hip")
    data_root, result_root = os.path.join(workdir, "data"), os.path.join(workdir, "output")
    init_path = os.path.join(workdir, "init.pth")
    if mode == "init":      # the shared starting point, written by the REFERENCE's classes
        torch.manual_seed(7)
        m = train_zju.MidasNet_small_videpth(device=torch.device("cpu"), min_pred=0.1, max_pred=255.0, in_channels=3)
        m.save(init_path)
        write_dataset(data_root, result_root)
        return
    ckpt = os.path.join(workdir, "ckpt_" + mode)
    settings = dict(      # the kwargs literal of train_zju.py:429-487 with the paths / step counts of this run
        train_root=data_root, scenes=["scene0"], image_file="thermal_undistort", mono_pred_file="any", radar_file="radar_png", gt_file="lidar_png_int",
        sparse_gt_file="lidar_png", result_root=result_root,
        learning_rates=[1e-4], learning_schedule=[2], batch_size=2, n_step_per_summary=10 ** 9, n_step_per_checkpoint=10 ** 9,
        random_crop_size=None, input_random_filp=False, input_random_brightness=None, input_random_contrast=None, input_random_saturation=None,
        input_random_radar_noise=None,
        loss_func='l1', w_smoothness=0.2, w_weight_decay=0.0, sobel_filter_size=7, w_lidar_loss=1.5, w_edge=0.0, w_unsupervised=0.0,
        ground_truth_outlier_removal_kernel_size=3, ground_truth_outlier_removal_threshold=1.5, ground_truth_dilation_kernel_size=-1,
        restore_path=init_path, min_pred=0.1, max_pred=255.0, min_depth=0.0, max_depth=100.0, checkpoint_dirpath=ckpt, n_threads=0,
        model_type='midas-small', interp='rcnet', random_rcnet_thr=None, global_alignment='s', mono_type='inv')
    # both frames form ONE batch and the run is two epochs = two optimisation steps: the DataLoader shuffles with a seed drawn from torch's global
    # generator, which the two model constructors advance differently -- with one frame per step the two variants would visit the frames in
    # different orders; the order inside a batch changes neither the BatchNorm statistics nor the loss
    losses = []
    real_print = print

    def spy(*a, **k):      # "<step>/<steps> epoch:<e>: <loss>" per step (train_zju.py:387)
        s = " ".join(str(x) for x in a)
        if " epoch:" in s:
            losses.append(float(s.rsplit(":", 1)[1]))
        real_print(*a, **k)
        sys.stdout.flush()
    train_zju.print = spy
    torch.manual_seed(11); np.random.seed(11); random.seed(11)
    train_zju.train(**settings)
    final = os.path.join(ckpt, "model-2.pth")      # (train_zju.py:421: the step count at the end of training)
    sd = torch.load(final, map_location="cpu")
    init = torch.load(init_path, map_location="cpu")
    keys = [k for k in sd if sd[k].is_floating_point() and "running" not in k]
    upd = torch.cat([(sd[k].double() - init[k].double()).reshape(-1) for k in keys])
    torch.save(upd, os.path.join(workdir, mode + "_update.pt"))
    # ---- val_zju.validate (val_zju.py:24-310) on the checkpoint train() just wrote, one frame -------------------------------------------------
    # module globals the script sets under __main__ (val_zju.py:313-323) and validate() reads (:81, :151, :166)
    val_zju.result_root, val_zju.min_pred, val_zju.max_pred = result_root, 0.1, 255.0
    model = val_zju.MidasNet_small_videpth(device=torch.device("cpu"), path=final, min_pred=0.1, max_pred=255.0, in_channels=3)
    model.eval()
    tf = val_zju.transforms.get_transforms(288, 288, depth_predictor='midas_small')
    best = dict(step=-1, mae=np.inf, rmse=np.inf, imae=np.inf, irmse=np.inf, abs_rel=np.inf, sq_rel=np.inf, delta1=0.0)
    # the frame count is the dataset's (2); the DataLoader of validate() uses one worker process (val_zju.py:113)
    res = val_zju.validate(best_results=best, ScaleMapLearner=model, step=2, ScaleMapLearner_transform=tf, min_depth_inference=0.0,
                           max_depth_inference=100.0, min_depth_val=0.0, max_depth_val=50.0, input_path=data_root, output_path=os.path.join(workdir, "val_" + mode),
                           scenes=["scene0"], save_output=False, log_path=os.path.join(workdir, "val_%s.txt" % mode), interp='rcnet', global_alignment='s',
                           mono_type='inv', mono_model='any')
    out = dict(mode=mode, losses=losses, checkpoint=sorted(os.listdir(ckpt)), n_keys=len(sd),
               val={k: float(v) for k, v in res.items()})
    json.dump(out, open(os.path.join(workdir, mode + ".json"), "w"))


def main():
    if len(sys.argv) == 3:
        return child(sys.argv[1], sys.argv[2])
    if not os.path.isdir(REF):
        sys.exit("run_reference_sml_unchanged.py needs /root/reference (build container only)")
    with tempfile.TemporaryDirectory() as wd:
        for mode in ("init", "ref", "hip"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), mode, wd], capture_output=True, text=True)
            tail = "\n".join(l for l in r.stdout.splitlines() if " epoch:" in l or "Begin training" in l or "MAE" in l or "e+" in l or "e-" in l)
            print("[%s] rc=%d\n%s" % (mode, r.returncode, tail), flush=True)
            if r.returncode != 0:
                sys.exit(r.stderr[-4000:])
        ref, hip = (json.load(open(os.path.join(wd, m + ".json"))) for m in ("ref", "hip"))
        import torch
        u_ref, u_hip = (torch.load(os.path.join(wd, m + "_update.pt")) for m in ("ref", "hip"))
        upd_cos = float(torch.dot(u_ref, u_hip) / (u_ref.norm() * u_hip.norm()))
    print("reference modules :", ref)
    print("riders_amd modules:", hip)
    assert len(ref["losses"]) == len(hip["losses"]) == 2 and ref["checkpoint"] == hip["checkpoint"] and ref["n_keys"] == hip["n_keys"]
    for a, b in zip(ref["losses"], hip["losses"]):
        assert abs(a - b) <= 1e-3 * abs(a), (a, b)
    print("two-step weight update (final - initial, %d elements): cosine %.6f" % (u_ref.numel(), upd_cos))
    assert upd_cos > 0.99, upd_cos
    d_abs = abs(ref["val"]["abs_rel"] - hip["val"]["abs_rel"])
    print("val_zju.validate abs_rel: reference modules %.6f, riders_amd modules %.6f (difference %.2e; north_star: within 1e-3)" % (
        ref["val"]["abs_rel"], hip["val"]["abs_rel"], d_abs))
    assert d_abs < 1e-3
    print("OK: the reference's train_zju.train() and val_zju.validate() ran unchanged on the aliased riders_amd modules")


if __name__ == "__main__":
    main()
