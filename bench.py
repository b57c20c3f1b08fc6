"""bench.py -- RIDERS training throughput on MI355X: BASELINE.json's metric "train imgs/sec (RC-Net+SML, 256x512)".

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Headline (`--workload chain`, the default): ONE timed loop whose iteration carries 16 synthetic 256x512 images through BOTH training stages --
two RC-Net optimisation steps (BASELINE configs[1]: batch 8, K = 30 radar points, patch 240x100, image edge-padded to 496x612) followed by one
Scale-Map-Learner optimisation step (configs[2]: batch 16) -- each step = normalise + labels / device pre-step + forward + loss + backward +
(RCCL gradient all-reduce for N > 1, started per stage while the backward is still running) + fused Adam, inputs resident in HBM, bf16.
`value` = 16 x K x N / elapsed, `ms_per_step` = one iteration.  The two stages are then timed on their own the same way (sub-objects `rcnet`,
`sml`); `--workload rcnet` / `sml` run one stage only (tools, profiler children).  Weak scaling: every rank processes its own images.

Timing: W warm-up iterations, then untimed "settling" replays until `--settle-seconds` of GPU work have passed (clocks and the SMI sampler
settle), then EXACTLY K timed iterations between barrier + synchronize pairs, max over ranks.

Launching: with WORLD_SIZE in the environment (torch.distributed.run) this process is one rank and `--gpus` must equal WORLD_SIZE (exit 2
otherwise).  Without it, `--gpus N` (N > 1) makes THIS process a GPU-free parent that starts N rank processes before anything touches a GPU,
relays rank 0's JSON line and exits with the worst rank's code.

Rank 0 prints ONE JSON line (< 6 KB; everything else goes to --full-json), including
  "roofline":      the dominant kernel FAMILY of the RC-Net step (all instantiations of one kernel template, e.g. conv3x3_frag_kernel): achieved =
                   algorithmic FLOPs (2 / MAC) or bytes (every operand once) of its launches / their summed durations.  Durations: HIP events
                   on the launch stream in instrumented eager steps right after the timed region (`avg_launch_us_hip_events`) AND the rocprofv3
                   --kernel-trace summary of the same workload collected by a child process of this run (`avg_launch_us`, the one `frac` uses:
                   the committed profiles/r06_*_kernel_stats.csv is that summary, so `frac` can be recomputed from it); `traffic` = HBM bytes
                   per launch from this run's own FETCH_SIZE / WRITE_SIZE counter passes (separate passes, gfx950 correction).
  "roofline_families": the same figures for the largest families (MFMA-bound convolutions, weight gradients, HBM-bound BatchNorm passes).
  "cpu_baseline":  the oracle (PyTorch-CPU restatement of the reference path, kind "port") timed on this box's host cores on a bounded sample.
  "unchanged_caller": the path an unmodified training script takes (RCNetModel.forward -> compute_loss -> loss.backward() ->
                   torch.optim.Adam.step() -> loss.item(); the SML twin): `value` with the forward region captured and replayed inside that loop
                   (engine.set_autograph, one line of INTEGRATION.md's aliasing block), `eager_value` with plain eager launches, next to the
                   fully captured step (`graphed_value`); launches per step from marker-cut kernel traces.
  "val_abs_rel":   the validation chain's abs-rel on 8 synthetic 256x512 frames: HIP fp32, HIP bf16 and the oracle on identical weights.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_HBM = 8000.0                                   # GB/s  (MI355X_MICROARCH.md)
PEAK_MFMA = {"fp32": 157.3, "bf16": 2500.0, "fp16": 2500.0}       # TFLOP/s dense
KSTATS_FILE = "profiles/r06_%s_kernel_stats.csv"                     # rocprofv3 --kernel-trace --stats summary of the same command
RIDGE = {k: v * 1e12 / (PEAK_HBM * 1e9) for k, v in PEAK_MFMA.items()}     # FLOP/B above which the MFMA roof binds
CHAIN_RC_STEPS = 2      # RC-Net steps (batch 8) per SML step (batch 16): the same 16 images pass both stages


# ================================================================================================ CPU baseline (oracle = test infrastructure)
def _rcnet_cpu_step(batch_n):
    """-> callable running one oracle RC-Net training step (fwd + loss + bwd + Adam) at batch `batch_n` (K=30, 256x512) on the CPU."""
    from oracle import rcnet as O
    from riders_amd import rcnet_main
    cfg = rcnet_main.ZJU_CONFIG
    torch.manual_seed(0)
    model = rcnet_main.build_model(torch.device('cpu'), cfg)
    sd_e = {k: v.detach().clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in model.encoder.state_dict().items()}
    sd_d = {k: v.detach().clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in model.decoder.state_dict().items()}
    leaves = [v for d in (sd_e, sd_d) for v in d.values() if v.requires_grad]
    state = [(torch.zeros_like(v), torch.zeros_like(v)) for v in leaves]
    img, pts, boxes, gt = rcnet_main.synthetic_batch(batch_n, 256, 512, cfg, seed=99)
    img = img / 255.0
    pts = pts.reshape(-1, 3)
    gt = gt.reshape(-1, 1, cfg['patch_size'][0], cfg['patch_size'][1])
    label, valid = O.rcnet_labels(gt, pts, 0.5)
    nstep = [0]

    def step():
        for v in leaves:
            v.grad = None
        logits = O.rcnet_forward(img, pts, [b for b in boxes], sd_e, sd_d, cfg['patch_size'], True)
        O.rcnet_loss(logits, label, valid, cfg['w_positive_class']).backward()
        nstep[0] += 1
        with torch.no_grad():
            for v, (m, s) in zip(leaves, state):
                if v.grad is not None:
                    p, m2, s2 = O.adam_step(v, v.grad, m, s, nstep[0], cfg['learning_rate'])
                    v.copy_(p); m.copy_(m2); s.copy_(s2)
    return step


def _sml_cpu_step(batch_n, h, w):
    """-> callable running one oracle SML training step (pre-step + fwd + loss + bwd + Adam via torch.optim) on the CPU."""
    import contextlib
    import numpy as np
    from oracle import sml as OS
    from riders_amd import sml_main
    cfg = sml_main.ZJU_SML_CONFIG
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = sml_main.build_model(torch.device('cpu'), cfg)
    o = OS.SMLOracle()
    o.load_state_dict(model.state_dict())
    o.train()
    opt = torch.optim.Adam(o.parameters(), lr=cfg['learning_rate'])
    image, mono, radar, gt, sparse_gt, rcnet = [b.numpy() for b in sml_main.synthetic_batch(batch_n, h, w, seed=98)]
    hw = sml_main.net_size(h, w)

    def step():
        xs, ds = [], []
        for i in range(batch_n):
            xo, do, _ = OS.prestep_sample(image[i], mono[i, 0], radar[i, 0], rcnet[i, 0], hw)
            xs.append(torch.from_numpy(np.ascontiguousarray(xo))); ds.append(torch.from_numpy(np.ascontiguousarray(do)))
        x, d = torch.stack(xs).float(), torch.stack(ds).float()
        gi = torch.stack([torch.from_numpy(np.ascontiguousarray(OS.nearest_resize(gt[i, 0], hw[0], hw[1]))) for i in range(batch_n)])[:, None].float()
        gs = torch.stack([torch.from_numpy(np.ascontiguousarray(OS.nearest_resize(sparse_gt[i, 0], hw[0], hw[1]))) for i in range(batch_n)])[:, None].float()
        gi = OS.remove_outliers(gi, cfg['outlier_removal_kernel_size'], cfg['outlier_removal_threshold'])
        opt.zero_grad()
        pred = o(x, d)
        loss, _ = OS.compute_loss(1.0 / d, 1.0 / pred, gi, gs, w_smoothness=cfg['w_smoothness'], sobel_filter_size=cfg['sobel_filter_size'],
                                  w_lidar_loss=cfg['w_lidar_loss'], w_edge=cfg['w_edge'])
        loss.backward()
        opt.step()
    return step


def _time_steps(step, warm, timed):
    for _ in range(warm):
        step()
    ts = []
    for _ in range(timed):
        t0 = time.time(); step(); ts.append(time.time() - t0)
    return min(ts)


def val_abs_rel_pair(dev, frames=8, H=256, W=512):
    """Validation chain (val_zju.py:124-254: device pre-step -> network (eval) -> 1/pred -> bicubic -> masked metrics) on identical random-init
    weights and `frames` synthetic HxW frames: abs-rel of the HIP path in fp32 (north_star: within 1e-3 of the reference path) and in bf16 (the
    driver-timed dtype), and of the oracle chain (fp32, CPU)."""
    import contextlib
    from oracle import sml as OS
    from riders_amd import engine, sml_main
    batch = sml_main.synthetic_batch(frames, H, W, seed=21)
    dbatch = tuple(b.to(dev) for b in batch)
    got, sd = {}, None
    for mode in ("fp32", "bf16"):
        engine.set_compute_dtype(mode); engine.clear_caches()
        try:
            torch.manual_seed(5)
            with contextlib.redirect_stdout(sys.stderr):
                m = sml_main.build_model(dev, sml_main.ZJU_SML_CONFIG)
            if sd is None:
                sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
            else:
                m.load_state_dict(sd)      # identical weights in both modes whatever the constructor drew
                engine.refresh_packed()
            m.eval()
            got[mode] = [float(v) for v in sml_main.validate_batch(m, dbatch)["abs_rel"]]
        finally:
            engine.set_compute_dtype("fp32"); engine.clear_caches()
    o = OS.SMLOracle()
    o.load_state_dict(sd)
    o.eval()
    image, mono, radar, gt, sparse_gt, rcnet = [b.numpy() for b in batch]
    hw = sml_main.net_size(H, W)
    ref = []
    for i in range(frames):
        xo, do, _ = OS.prestep_sample(image[i], mono[i, 0], radar[i, 0], rcnet[i, 0], hw)
        with torch.no_grad():
            po = o(torch.from_numpy(xo)[None].float(), torch.from_numpy(do)[None].float())
        ref.append(float(OS.val_metrics(po, sparse_gt[i, 0], (H, W))["abs_rel"]))
    mean = lambda v: sum(v) / len(v)      # noqa: E731
    out = dict(oracle=mean(ref), sample="%d synthetic %dx%d frames (network input %dx%d), random-init weights identical on every path" % (frames, H, W, hw[0], hw[1]))
    for mode in ("fp32", "bf16"):
        out[mode] = dict(hip=mean(got[mode]), diff_of_means=abs(mean(got[mode]) - mean(ref)), max_abs_diff=max(abs(a - b) for a, b in zip(got[mode], ref)))
    # (round 5's keys, fp32 figures)
    out["hip"], out["max_abs_diff"] = out["fp32"]["hip"], out["fp32"]["max_abs_diff"]
    return out


def cpu_baseline(budget_s=12.0):
    """Oracle training steps with torch CPU ops on the host cores, bounded to ~budget_s seconds: thread sweep (8, 16, 32, 64) on the RC-Net
    B = 1 step, then RC-Net B = 8 and the SML step at the best thread count; imgs/s of the best configuration is `value`."""
    t_start = time.time()
    ncpu = os.cpu_count() or 1
    # 8..64 threads: the oracle's small convolutions stop scaling long before a 128/256-thread host is full (round 3: 1.68 img/s at 16
    # threads, 0.59 at 64, 0.003 at 256 -- one such trial took 17 minutes), so "all cores" is not a candidate and the sweep stops as soon
    # as a count is clearly slower than the best so far
    counts = sorted({c for c in (8, 16, 32, 64) if c <= ncpu} or {ncpu})
    keep = torch.get_num_threads()
    step1 = _rcnet_cpu_step(1)
    sweep = {}
    for c in counts:
        torch.set_num_threads(c)
        sweep[c] = 1.0 / _time_steps(step1, 1, 2)
        if sweep[c] < 0.8 * max(sweep.values()) or time.time() - t_start > 0.5 * budget_s:
            break
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    out = dict(value=sweep[best], unit="imgs/s", cores=best, kind="port", host_cpus=ncpu,
               thread_sweep_b1={str(k): v for k, v in sweep.items()},
               sample="oracle RC-Net full step (fwd+loss+bwd+Adam), B=1 (30 ROIs, 256x512), fp32, best of 2 timed steps after 1 warm-up per thread count")
    if time.time() - t_start < 0.6 * budget_s:
        out["rcnet_b8"] = dict(value=8.0 / _time_steps(_rcnet_cpu_step(8), 0, 1), unit="imgs/s", cores=best, sample="one B=8 step (240 ROIs), no warm-up")
        out["value"] = max(out["value"], out["rcnet_b8"]["value"])
    if time.time() - t_start < 0.85 * budget_s:
        try:
            out["sml_b1"] = dict(value=1.0 / _time_steps(_sml_cpu_step(1, 256, 512), 0, 1), unit="imgs/s", cores=best,
                                 sample="oracle SML step (pre-step+fwd+loss+bwd+Adam), B=1, 256x512 frame, one timed step, no warm-up")
            # BASELINE's metric names both stages: images/s through an RC-Net step and an SML step on the host cores
            out["chained"] = dict(value=1.0 / (1.0 / out["value"] + 1.0 / out["sml_b1"]["value"]), unit="imgs/s",
                                  sample="per-image time = oracle RC-Net step + oracle SML step (the two figures above)")
        except Exception as ex:      # the baseline leg never fails the bench line
            out["sml_b1"] = dict(error=repr(ex)[:200])
    torch.set_num_threads(keep)
    out["seconds"] = time.time() - t_start
    return out


# ================================================================================================ roofline objects
def family_of(kernel_name):
    """kernel template base name = the family: 'conv3x3_frag_kernel<rd::bf16_t, 4, 2, 4, ...>' -> 'conv3x3_frag_kernel'"""
    return kernel_name.split("<")[0].split("(")[0].strip()


def family_table(timer, timed_steps, ms_per_step, dtype):
    """Per-engine-kind table from the HIP-event records (kind = what the engine tags a launch with: conv_gemm, conv_wgrad, bn_backward, ...)."""
    peak_mfma = PEAK_MFMA[dtype]
    fams = {}
    for (kind, desc), (n, tms, fl, by) in timer.detail().items():
        f = fams.setdefault(kind, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, t_roof_ms=0.0, t_mfma_ms=0.0, t_hbm_ms=0.0))
        t_m, t_h = fl / (peak_mfma * 1e12) * 1e3, by / (PEAK_HBM * 1e9) * 1e3
        f["ms"] += tms; f["flops"] += fl; f["bytes"] += by; f["launches"] += n
        f["t_roof_ms"] += max(t_m, t_h); f["t_mfma_ms"] += t_m if t_m >= t_h else 0.0; f["t_hbm_ms"] += t_h if t_h > t_m else 0.0
    table = {}
    for k, f in fams.items():
        bound = "mfma" if f["t_mfma_ms"] >= f["t_hbm_ms"] else "hbm"
        ach = f["flops"] / (f["ms"] * 1e-3) / 1e12 if bound == "mfma" else f["bytes"] / (f["ms"] * 1e-3) / 1e9
        peak = peak_mfma if bound == "mfma" else PEAK_HBM
        table[k] = dict(ms_per_step=f["ms"] / timed_steps, share_of_step=(f["ms"] / timed_steps) / ms_per_step, launches_per_step=f["launches"] / timed_steps,
                        bound=bound, achieved=ach, peak=peak, unit="TFLOP/s" if bound == "mfma" else "GB/s", frac=ach / peak,
                        frac_time_weighted=f["t_roof_ms"] / f["ms"] if f["ms"] > 0 else 0.0)
    return table


def kernel_tables(timer, timed_steps, dtype):
    """(per instantiation, per family) dicts from the instrumented steps: launches / ms (HIP events) / algorithmic flops / bytes PER STEP."""
    inst, fam = {}, {}
    for name, v in timer.by_kernel().items():
        e = dict(launches=v["launches"] / timed_steps, ms=v["ms"] / timed_steps, flops=v["flops"] / timed_steps, bytes=v["bytes"] / timed_steps, kind=v.get("kind"))
        inst[name] = e
        f = fam.setdefault(family_of(name), dict(launches=0.0, ms=0.0, flops=0.0, bytes=0.0, instantiations=0))
        for k in ("launches", "ms", "flops", "bytes"):
            f[k] += e[k]
        f["instantiations"] += 1
    return inst, fam


def roofline_entry(name, work, dtype, ms_per_step, rocprof=None, live=None, what="kernel family"):
    """One roofline object.  work: dict(launches, ms, flops, bytes) per step from the instrumented steps (HIP events); rocprof: optional
    dict(calls_per_step, ns_per_step) from the kernel-trace child (then ITS durations make `achieved` / `frac`); live: optional counter figures
    dict(bytes_per_launch, dispatches[, mfma_util])."""
    peak_mfma = PEAK_MFMA[dtype]
    intensity = work["flops"] / max(work["bytes"], 1.0)
    bound = "mfma" if intensity >= RIDGE[dtype] else "hbm"
    ms_ev = work["ms"]
    use_prof = bool(rocprof and rocprof.get("ns_per_step", 0) > 0)
    ms = rocprof["ns_per_step"] * 1e-6 if use_prof else ms_ev
    launches = rocprof["calls_per_step"] if use_prof else work["launches"]
    tot = work["flops"] if bound == "mfma" else work["bytes"]
    ach = tot / (ms * 1e-3) / (1e12 if bound == "mfma" else 1e9) if ms > 0 else 0.0
    peak = peak_mfma if bound == "mfma" else PEAK_HBM
    r = dict(bound=bound, kernel=name, what=what, achieved=ach, peak=peak, unit="TFLOP/s" if bound == "mfma" else "GB/s", frac=ach / peak, traffic=None,
             launches_per_step=launches, avg_launch_us=ms * 1e3 / max(launches, 1e-9), avg_launch_us_hip_events=ms_ev * 1e3 / max(work["launches"], 1e-9),
             ms_per_step=ms, share_of_step=ms / ms_per_step if ms_per_step > 0 else 0.0,
             algorithmic_flops_per_launch=work["flops"] / max(work["launches"], 1e-9), algorithmic_bytes_per_launch=work["bytes"] / max(work["launches"], 1e-9),
             flops_per_byte=intensity, instantiations=work.get("instantiations", 1),
             duration_source=("rocprofv3 --kernel-trace --stats of the same captured step (hipGraph replays), collected by a child process of this run" if use_prof
                              else "HIP events on the launch stream, instrumented eager steps after the timed region"))
    if use_prof and abs(rocprof["calls_per_step"] - work["launches"]) > 0.01 * max(work["launches"], 1.0):
        r["launch_count_mismatch"] = dict(rocprof=rocprof["calls_per_step"], hip_events=work["launches"])
    if live:
        r["traffic"] = live["bytes_per_launch"]
        r["traffic_over_algorithmic"] = live["bytes_per_launch"] / max(r["algorithmic_bytes_per_launch"], 1.0)
        r["traffic_live"] = True
        r["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (--kernel-trace only, separate passes) of the same workload, started as child "
                               "processes by THIS run: mean over %d dispatches; FETCH_SIZE doubled (gfx950)" % live["dispatches"])
        if "mfma_util" in live:
            r["mfma_util"] = live["mfma_util"]
    return r


def build_rooflines(timer, timed_steps, ms_per_step, dtype, prof=None, live=None, key=""):
    """-> dict(roofline = dominant family, roofline_families = the largest families, roofline_kernel = dominant single instantiation, tables).
    prof: {kernel instantiation name: (calls per step, ns per step)} from the kernel-trace child; live: {instantiation: counter figures}."""
    inst, fam = kernel_tables(timer, timed_steps, dtype)
    if not fam:
        return dict(roofline=None, families=family_table(timer, timed_steps, ms_per_step, dtype))
    prof_f, live_f = {}, {}
    # (only the instantiations whose algorithmic work the engine tallied: a template's other uses -- e.g. col_reduce_vec_kernel<.., 1, ..> as the
    # bias column sum -- carry time but no tallied work and would dilute the family)
    for n, (calls, ns) in (prof or {}).items():
        if n not in inst:
            continue
        e = prof_f.setdefault(family_of(n), dict(calls_per_step=0.0, ns_per_step=0.0))
        e["calls_per_step"] += calls; e["ns_per_step"] += ns
    for n, t in (live or {}).items():
        if n not in inst:
            continue
        e = live_f.setdefault(family_of(n), dict(bytes=0.0, dispatches=0, busy=0.0, act=0.0))
        e["bytes"] += t["bytes_per_launch"] * t["dispatches"]; e["dispatches"] += t["dispatches"]
        if "mfma_busy_cycles" in t:
            e["busy"] += t["mfma_busy_cycles"] * t["dispatches"]; e["act"] += t["gui_active"] * t["dispatches"]
    for e in live_f.values():
        e["bytes_per_launch"] = e["bytes"] / max(e["dispatches"], 1)
        if e["act"] > 0:
            e["mfma_util"] = e["busy"] / (1024.0 * e["act"] / 8.0)
    entries = {}
    for name, w in fam.items():
        entries[name] = roofline_entry(name, w, dtype, ms_per_step, prof_f.get(name), live_f.get(name))
        entries[name]["instantiation_names"] = sorted(n for n in inst if family_of(n) == name)
    order = sorted(entries, key=lambda k: -entries[k]["ms_per_step"])
    dom = entries[order[0]]
    dom["note"] = ("dominant = the kernel family (all instantiations of one template) with the largest summed launch time per RC-Net / SML step; the rocprofv3 "
                   "--kernel-trace --stats summary of the same command is " + KSTATS_FILE % key)
    dk = max(inst, key=lambda k: inst[k]["ms"])
    p1 = (prof or {}).get(dk)
    rk = roofline_entry(dk, inst[dk], dtype, ms_per_step, dict(calls_per_step=p1[0], ns_per_step=p1[1]) if p1 else None,
                        (live or {}).get(dk), what="kernel instantiation")
    return dict(roofline=dom, roofline_families=[entries[k] for k in order[:6]], roofline_kernel=rk,
                kernels={n: dict(ms_per_step=v["ms"], launches_per_step=v["launches"], tflops=v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0,
                                 gbs=v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0) for n, v in sorted(inst.items(), key=lambda kv: -kv[1]["ms"])[:12]},
                families=family_table(timer, timed_steps, ms_per_step, dtype))


# ================================================================================================ profiler children
_PROFILER_VARS = ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY", "ROCPROF_", "ROCPROFILER_", "ROCP_", "HSA_TOOLS_LIB", "ROCTX_")
MARKER = "spin_kernel"      # torch.cuda._sleep's kernel: the children bracket their steps with it so that dispatches can be counted per step


def under_profiler(env=None):
    """True when this process was started by rocprofv3 / rocprof (their tool library is preloaded or named in the environment)."""
    env = os.environ if env is None else env
    if any(("rocprof" in v.lower() or "roctracer" in v.lower()) for v in (env.get("LD_PRELOAD", ""), env.get("HSA_TOOLS_LIB", ""))):
        return True
    return any(k.startswith(_PROFILER_VARS) for k in env)


def clean_profiler_env(env):
    """a copy of `env` without any profiler's variables (children that run their OWN rocprofv3 must not inherit an outer one's)"""
    out = {k: v for k, v in env.items() if not k.startswith(_PROFILER_VARS)}
    if "LD_PRELOAD" in out:
        keep = [x for x in out["LD_PRELOAD"].replace(":", " ").split() if "rocprof" not in x.lower() and "roctracer" not in x.lower()]
        if keep:
            out["LD_PRELOAD"] = ":".join(keep)
        else:
            del out["LD_PRELOAD"]
    return out


def _child_base(args, workload, outdir, steps=2, warmup=1, eager=True, settle=0.0):
    base = ["python3", os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup", str(warmup), "--settle-seconds", str(settle),
            "--no-cpu-baseline", "--no-legs", "--no-children", "--batch", str(args.batch), "--dtype", args.dtype, "--height", str(args.height),
            "--width", str(args.width), "--sml-batch", str(args.sml_batch), "--sml-height", str(args.sml_height), "--sml-width", str(args.sml_width),
            "--full-json", os.path.join(outdir, "child_full.json")]
    if eager:
        base.append("--eager")
    if getattr(args, "autograph", False):
        base.append("--autograph")
    if args.opts:
        base += ["--opts", args.opts]
    return base


CHILD_BUDGET = {"left": 300.0}      # seconds all profiler children of one run may take together (normally ~50 s); beyond it the rest are skipped


def _run_child(cmd, budget_s):
    import signal
    import subprocess
    budget_s = min(budget_s, CHILD_BUDGET["left"])
    if budget_s < 20.0:
        sys.stderr.write("bench.py: the profiler children's time budget is spent -- skipping one\n")
        return -2
    t0 = time.time()
    try:
        return _run_child_inner(cmd, budget_s, signal, subprocess)
    finally:
        CHILD_BUDGET["left"] -= time.time() - t0


def _run_child_inner(cmd, budget_s, signal, subprocess):
    pr = subprocess.Popen(cmd, cwd="/tmp", env=dict(clean_profiler_env(os.environ), TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                          start_new_session=True)
    try:
        return pr.wait(timeout=budget_s)
    except subprocess.TimeoutExpired:
        os.killpg(pr.pid, signal.SIGKILL)
        pr.wait()
        return -1


def _children_possible():
    import shutil
    if shutil.which("rocprofv3") is None:
        return False
    if under_profiler():      # this run is itself being profiled: a child profiler would be a profiler inside a profiler
        sys.stderr.write("bench.py: running under a profiler -- the profiler children (kernel trace, counter passes) are skipped\n")
        return False
    return True


def kernel_trace_child(args, workload, budget_s=120.0, steps=16, save_stats=None):
    """rocprofv3 --kernel-trace --stats of `workload` in a CHILD process (this process holds the GPU and never replaces itself): the stage's
    CAPTURED step, replayed `steps` times after a second of settling -- the very launches the timed region runs (rocprofv3 traces the kernels of
    a hipGraph replay one by one); the unchanged-caller workloads run eagerly, as they do.  The child brackets every step with a marker kernel:
    the dispatches between two markers are ONE step, and the first steps (allocation, clocks still ramping) are left out of the averages.
    -> dict(per_kernel={instantiation: (calls per step, ns per step)}, launches_per_step, kernel_ms_per_step, wall_ms_per_step) or None."""
    import csv
    import glob
    import shutil
    import tempfile
    if not _children_possible():
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from traffic_aggregate import kernel_key
    out = tempfile.mkdtemp(prefix="riders_ktrace_", dir="/tmp")
    try:
        caller = workload.startswith("caller")
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(out, "t"), "-o", "p", "--"] + \
            _child_base(args, workload, out, steps=steps, warmup=2, eager=caller, settle=0.0 if caller else 1.0) + ["--step-markers"]
        if _run_child(cmd, budget_s) != 0:
            return None
        rows = []
        for f in glob.glob(os.path.join(out, "t", "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f, newline="")):
                rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
        rows.sort()
        marks = [i for i, r in enumerate(rows) if MARKER in r[2]]
        if len(marks) < 2:
            sys.stderr.write("bench.py: kernel-trace child: no step markers in the trace (%d dispatches)\n" % len(rows))
            return None
        # the complete steps between two markers, without the first ones (allocation, clock ramp): mean over the rest
        spans = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1)]
        spans = spans[min(4, len(spans) - 1):]
        per, nl, kms, wall = {}, 0, 0.0, 0.0
        for a, b in spans:
            wall += (rows[b][0] - rows[a][1]) * 1e-6      # end of the opening marker .. start of the closing one
            for s, e, name in rows[a + 1:b]:
                nl += 1
                kms += (e - s) * 1e-6
                if "rd::" in name or "rd_f16::" in name:
                    k = kernel_key(name)
                    c = per.setdefault(k, [0, 0.0])
                    c[0] += 1; c[1] += (e - s)
        n = float(len(spans))
        if save_stats:
            for f in glob.glob(os.path.join(out, "t", "**", "*kernel_stats.csv"), recursive=True):
                try:
                    os.makedirs(os.path.dirname(os.path.abspath(save_stats)), exist_ok=True)
                    shutil.copy(f, save_stats)
                except OSError:
                    pass
        return dict(per_kernel={k: (c[0] / n, c[1] / n) for k, c in per.items()}, launches_per_step=nl / n, kernel_ms_per_step=kms / n, wall_ms_per_step=wall / n,
                    steps_averaged=int(n), mode="eager" if caller else "captured step (hipGraph replays)")
    except Exception as ex:      # never lose the bench line to an auxiliary measurement
        sys.stderr.write("bench.py: kernel-trace child failed (%r)\n" % (ex,))
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def live_traffic(args, workload="rcnet", budget_s=45.0):
    """HBM bytes per launch and matrix-pipe busy fraction of every rd:: kernel, measured by THIS run: three rocprofv3 counter passes (FETCH_SIZE;
    WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE -- separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes) of the same
    workload as child processes (3 eager steps each), with the units / gfx950 correction of tools/traffic_aggregate.py.
    -> {kernel instantiation: dict(bytes_per_launch, dispatches[, mfma_util, mfma_busy_cycles, gui_active])} or None."""
    import csv
    import glob
    import shutil
    import tempfile
    if not _children_possible():
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from traffic_aggregate import kernel_key
    out = tempfile.mkdtemp(prefix="riders_live_traffic_", dir="/tmp")
    per = {}
    try:
        for n, ctrs in enumerate((("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))):
            d = os.path.join(out, "pass%d" % n)
            cmd = ["rocprofv3", "--pmc"] + list(ctrs) + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + _child_base(args, workload, out)
            rc = _run_child(cmd, budget_s)
            found = False
            if rc != 0 and n < 2:      # a failed / timed-out traffic pass: do not spend another budget on the next one
                return None
            if rc == 0:
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f, newline="")):
                        if row["Counter_Name"] not in ctrs or "rd" not in row["Kernel_Name"]:
                            continue
                        e = per.setdefault(kernel_key(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], [0.0, 0])
                        e[0] += float(row["Counter_Value"]); e[1] += 1
                        found = True
            if not found and n < 2:      # the traffic passes are the point; the matrix-pipe pass is an extra
                return None
    except Exception as ex:
        sys.stderr.write("bench.py: live counter pass failed (%r)\n" % (ex,))
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)
    res = {}
    for k, v in per.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            m = {c: x[0] / x[1] for c, x in v.items()}
            # KB; gfx950: wide streaming reads are tallied at half their size (MI355X_MICROARCH.md, HBM section)
            res[k] = dict(bytes_per_launch=m["FETCH_SIZE"] * 2.0 * 1024.0 + m["WRITE_SIZE"] * 1024.0, dispatches=v["FETCH_SIZE"][1])
            if m.get("GRBM_GUI_ACTIVE"):      # busy cycles summed over the 1024 SIMDs / (1024 x the dispatch's cycles; GRBM_GUI_ACTIVE is summed over 8 XCDs)
                res[k]["mfma_busy_cycles"], res[k]["gui_active"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), m["GRBM_GUI_ACTIVE"]
                res[k]["mfma_util"] = res[k]["mfma_busy_cycles"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
    return res or None


# ================================================================================================ the one stdout line
LINE_LIMIT = 6000      # bytes of the ONE stdout line (the driver kept ~8 KB of stdout and lost the head of a 39.8-KB line)
FLAT_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "launches_per_step",
                  "avg_launch_us", "avg_launch_us_hip_events", "ms_per_step", "share_of_step", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch",
                  "mfma_util", "traffic_live", "instantiations")


def _sig(v, n=5):
    """floats to n significant digits (the line is for a parser with a size limit, not for arithmetic)"""
    if isinstance(v, float):
        return float("%.*g" % (n, v)) if v == v and abs(v) != float("inf") else None
    return v


def flat_roofline(roof, keys=FLAT_ROOF_KEYS):
    """the flat part of a roofline object: no tables, no notes"""
    if not roof or "kernel" not in roof:
        return None
    return {k: _sig(roof[k]) for k in keys if k in roof}


LEG_METRIC = {"fp32": "train imgs/sec (RC-Net, batch 8, 256x512, fp32: the 1e-3 parity mode)",
              "config4": "train imgs/sec (RC-Net, batch 8 per GPU, 3x512x1024, fp16: BASELINE configs[4] per rank)",
              "config4_sml": "train imgs/sec (Scale Map Learner, batch 8 per GPU, 512x1024 at native resolution, fp16: BASELINE configs[4] per rank)"}
LEG_DTYPE = {"fp32": "f32", "config4": "f16", "config4_sml": "f16"}
METRIC = {"chain": "train imgs/sec (RC-Net+SML, 256x512)",
          "rcnet": "train imgs/sec (RC-Net, 256x512 thermal + 30 radar points, patch 240x100)",
          "sml": "train imgs/sec (Scale Map Learner, MiDaS-small / EfficientNet-Lite3)",
          "caller_rcnet": "train imgs/sec (RC-Net through the unchanged caller: eager, torch.optim.Adam, loss.item() per step)",
          "caller_sml": "train imgs/sec (SML through the unchanged caller: eager, torch.optim.Adam, loss.item() per step)"}


def leg_workload(kind, leg):
    if kind == "sml":
        return "SML training step, batch %d/GPU, %dx%d frames, device pre-step+fwd+loss+bwd+Adam" % (leg["batch_per_gpu"], leg["height"], leg["width"])
    return "RC-Net training step, batch %d/GPU, ZJU config (K=30, patch 240x100), %dx%d image, fwd+loss+bwd+Adam" % (leg["batch_per_gpu"], leg["height"], leg["width"])


def leg_record(kind, leg, dtype):
    r = {"metric": METRIC[kind], "value": leg["value"], "unit": "imgs/s", "ms_per_step": leg["ms_per_step"], "steps": leg["steps"], "dtype": dtype,
         "config": {"workload": leg_workload(kind, leg)}, "final_loss": leg["final_loss"], "settle_steps": leg.get("settle_steps"),
         "launch_mode": leg.get("launch_mode")}
    for k in ("roofline", "roofline_families", "roofline_kernel", "kernels", "families", "launches_per_step", "kernel_ms_per_step", "traced_wall_ms_per_step"):
        if leg.get(k) is not None:
            r[k] = leg[k]
    return r


def full_record(args, world, comm, ddp, head, rc, sml, legs, cpu, val, caller=None):
    """Everything the run measured: written to the --full-json file.  head: the headline leg (the chain at the default workload, else the one
    stage that was asked for); rc / sml: the stages timed on their own (None when not run)."""
    dt = {"fp32": "f32", "bf16": "bf16", "fp16": "f16"}[args.dtype]
    wl = args.workload
    if wl == "chain":
        per = head["images_per_step"]
        workload = ("%d RC-Net steps (batch %d, K=30, patch 240x100, %dx%d) + 1 SML step (batch %d, %dx%d) per iteration = %d images through BOTH stages; "
                    "each step fwd+loss+bwd+Adam" % (CHAIN_RC_STEPS, rc["batch_per_gpu"], rc["height"], rc["width"], sml["batch_per_gpu"], sml["height"], sml["width"], per))
    else:
        per = head["batch_per_gpu"]
        workload = leg_workload("sml" if wl.endswith("sml") else "rcnet", head) + (" (unchanged caller: eager, torch.optim.Adam)" if wl.startswith("caller") else "")
    src = rc if (wl == "chain" and rc is not None) else head
    out = {
        "metric": METRIC[wl], "value": head["value"], "unit": "imgs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dt, "data": "synthetic",
        "config": {"workload": workload, "global_batch": per * world, "parallelism": "dp%d" % world,
                   "note": ("BASELINE configs[3]: global batch 32 over %d rank(s)" % world) if args.config3 else
                           "BASELINE configs[1] + configs[2] per rank at every N (weak scaling); configs[3]'s global 32 on 8 GPUs is --config3"},
        "final_loss": head["final_loss"], "launch_mode": head.get("launch_mode"), "settle_steps": head.get("settle_steps"),
        "world_size": world, "comm": comm,
        "allreduce": None if not ddp else "RCCL sum (%s, %s) of the flat fp32 gradient arena in 3 stage buckets per model, each started when the backward "
                                          "passes its stage mark (overlaps the remaining backward); 1/N folded into Adam" % (
                                              args.allreduce, "rd_allreduce_bucket captured in the step graph" if getattr(args, "rccl_comm", None) is not None
                                              else "torch.distributed between stage graphs"),
        "roofline": src.get("roofline"), "roofline_families": src.get("roofline_families"), "roofline_kernel": src.get("roofline_kernel"),
        "roofline_workload": "the RC-Net step" if (wl == "chain" or wl.endswith("rcnet")) else "the SML step",
    }
    if wl == "chain":
        out["rcnet"] = leg_record("rcnet", rc, dt)
        out["sml"] = leg_record("sml", sml, dt)
        out["chain_vs_parts"] = {"chain_ms": head["ms_per_step"], "parts_ms": CHAIN_RC_STEPS * rc["ms_per_step"] + sml["ms_per_step"]}
    else:
        for k in ("kernels", "families", "launches_per_step", "kernel_ms_per_step"):
            if head.get(k) is not None:
                out[k] = head[k]
    for name, leg in legs.items():
        r = leg_record("sml" if name.endswith("_sml") else "rcnet", leg, LEG_DTYPE[name])
        r["metric"] = LEG_METRIC[name]
        out[name] = r
    if caller:
        out["unchanged_caller"] = caller
    if val is not None:
        out["val_abs_rel"] = val
    if cpu is not None:
        out["cpu_baseline"] = cpu
    return out


def compact_line(full):
    """The ONE stdout line: headline fields, flat `roofline` + the largest families, `cpu_baseline` (value, unit, cores, kind, sample), the stages /
    secondary legs as value / ms_per_step / dtype / config.workload / roofline.{kernel, bound, frac}.  Always < LINE_LIMIT bytes."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _sig(full[k], 7) for k in keep}
    cfg = full["config"]
    line["config"] = {"workload": cfg["workload"], "global_batch": cfg["global_batch"], "parallelism": cfg["parallelism"]}
    line["roofline"] = flat_roofline(full.get("roofline"))
    if line["roofline"] is not None:
        line["roofline"]["workload"] = full.get("roofline_workload")
    fams = full.get("roofline_families") or []
    if fams:
        line["roofline_families"] = [{k: _sig(f[k], 4) for k in ("kernel", "bound", "frac", "achieved", "unit", "share_of_step", "avg_launch_us", "launches_per_step",
                                                                    "traffic_over_algorithmic", "mfma_util") if f.get(k) is not None} for f in fams[:4]]
    cb = full.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = {k: _sig(cb[k]) for k in ("value", "unit", "cores", "kind", "sample", "host_cpus") if k in cb}
        if isinstance(cb.get("chained"), dict) and "value" in cb["chained"]:
            line["cpu_baseline"]["chained_value"] = _sig(cb["chained"]["value"])
    for name in ("rcnet", "sml", "fp32", "config4", "config4_sml"):
        leg = full.get(name)
        if leg is None:
            continue
        r = leg.get("roofline") or {}
        line[name] = {"value": _sig(leg["value"], 7), "unit": leg.get("unit", "imgs/s"), "ms_per_step": _sig(leg["ms_per_step"], 7), "dtype": leg["dtype"],
                      "config": {"workload": leg["config"]["workload"]},
                      "roofline": {k: _sig(r[k]) for k in ("kernel", "bound", "frac") if k in r}}
        if leg.get("launches_per_step") is not None:
            line[name]["launches_per_step"] = _sig(leg["launches_per_step"], 5)
    uc = full.get("unchanged_caller")
    if uc:
        line["unchanged_caller"] = {k: {a: _sig(v[a], 6) for a in ("value", "ms_per_step", "launches_per_step", "eager_value", "eager_ms_per_step", "eager_launches_per_step",
                                                                      "graphed_value") if v.get(a) is not None}
                                    for k, v in uc.items() if isinstance(v, dict)}
    va = full.get("val_abs_rel")
    if isinstance(va, dict) and "oracle" in va:
        line["val_abs_rel"] = {"oracle": _sig(va["oracle"], 6), "frames": va.get("sample", "")[:40]}
        for m in ("fp32", "bf16"):
            if isinstance(va.get(m), dict):
                line["val_abs_rel"][m] = {k: _sig(va[m][k], 6) for k in ("hip", "max_abs_diff")}
    for k in ("final_loss", "launch_mode", "world_size"):
        line[k] = _sig(full.get(k), 7)
    if full.get("comm"):
        line["comm"] = full["comm"]
    return line


def render_line(line, limit=LINE_LIMIT):
    """json text of the compact line; should a field ever grow past the limit, optional blocks are dropped (last first) rather than the line lost"""
    text = json.dumps(line, separators=(",", ":"))
    for k in ("comm", "launch_mode", "val_abs_rel", "unchanged_caller", "config4_sml", "config4", "fp32", "roofline_families", "sml", "rcnet"):
        if len(text) < limit:
            break
        line = {a: b for a, b in line.items() if a != k}
        text = json.dumps(line, separators=(",", ":"))
    return text


# ================================================================================================ the measured legs
class Leg(object):
    """One model + optimizer (+ reducer) + its static synthetic batch + the step callable (a captured hipGraph step unless --eager)."""
    pass


def build_leg(kind, args, dev, world, rank, **override):
    """override: dtype / batch / height / width / ... of a secondary leg (the fp32 parity mode, configs[4]) on a copy of args."""
    from riders_amd import engine, rcnet_main, sml_main
    if override:
        args = argparse.Namespace(**dict(vars(args), **override))
    from riders_amd.optim import FlatAdam
    from riders_amd.parallel import GradientAllReducer, rcnet_stages, sml_stages
    engine.set_compute_dtype(args.dtype)
    torch.manual_seed(0)  # identical initial weights on every rank
    L = Leg()
    L.kind, L.args, L.override = kind, args, override
    if kind == "sml":
        L.batch_n, L.h, L.w = args.sml_batch, args.sml_height, args.sml_width
        cfg = dict(sml_main.ZJU_SML_CONFIG, net_hw=(L.h, L.w)) if getattr(args, "sml_native", False) else sml_main.ZJU_SML_CONFIG
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):      # the constructor prints like the reference's; stdout carries the ONE JSON line only
            model = sml_main.build_model(dev, cfg)
        main_mod, extra, stages = sml_main, dict(outlier=sml_main.make_outlier_removal(cfg)), sml_stages(model)
        batch = sml_main.synthetic_batch(L.batch_n, L.h, L.w, seed=1234 + rank, device=dev)
    else:
        L.batch_n, L.h, L.w = args.batch, args.height, args.width
        cfg = rcnet_main.ZJU_CONFIG
        model = rcnet_main.build_model(dev, cfg)
        main_mod, extra, stages = rcnet_main, {}, rcnet_stages(model)
        batch = rcnet_main.synthetic_batch(L.batch_n, L.h, L.w, cfg, seed=1234 + rank, device=dev)
    model.train()
    opt = FlatAdam(model.parameters(), lr=cfg['learning_rate'])
    reducer = GradientAllReducer(opt, stages=stages, mode=args.allreduce, comm=args.rccl_comm) if (world > 1 or args.force_ddp) else None
    if reducer is not None:
        reducer.broadcast_parameters(0)
    extra["loss_scale"] = args.loss_scale if args.loss_scale is not None else (16384.0 if args.dtype == "fp16" else 1.0)
    L.model, L.opt, L.reducer, L.batch, L.cfg, L.extra, L.main_mod = model, opt, reducer, batch, cfg, extra, main_mod

    def eager_step():
        return main_mod.train_step(model, opt, batch, cfg, L.reducer, **extra)
    L.eager_step = eager_step
    if args.eager:
        L.step = eager_step
    else:
        try:
            L.step = main_mod.GraphedTrainStep(model, opt, batch, cfg, reducer, **extra)
        except Exception as ex:
            if reducer is None or reducer.comm is None:
                raise
            # the captured-collective step could not be built (every rank runs the same code on the same shapes, so every rank lands here):
            # fall back to torch.distributed's collectives between per-stage graphs rather than lose the measurement
            sys.stderr.write("bench.py: rank %d: single-graph step with captured RCCL buckets failed (%r); falling back to --comm torch\n" % (rank, ex))
            torch.cuda.synchronize()
            reducer.close()
            L.reducer = reducer = GradientAllReducer(opt, stages=stages, mode=args.allreduce)
            L.step = main_mod.GraphedTrainStep(model, opt, batch, cfg, reducer, **extra)
    L.launch_mode = "eager" if args.eager else (
        "one hipGraph (fwd+bwd+captured RCCL buckets) + eager Adam" if (L.reducer is not None and L.reducer.comm is not None) else
        "hipGraphs split at the stage marks (fwd+bwd) + eager all-reduce/Adam" if L.reducer is not None else "one hipGraph (fwd+bwd) + eager Adam")
    return L


def close_leg(L):
    from riders_amd import engine
    if L.reducer is not None:
        L.reducer.close()
    L.step = L.eager_step = L.model = L.opt = L.reducer = None
    engine.clear_caches()
    torch.cuda.empty_cache()


def _barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def settle(fn, seconds, world, dev):
    """untimed replays of fn until `seconds` of wall time have passed (rank 0 decides, so that every rank runs the same number) -> count"""
    n = 0
    if seconds <= 0:
        return 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go = True
    while go:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        n += 5
        go = time.perf_counter() - t0 < seconds
        if world > 1:
            import torch.distributed as dist
            flag = torch.tensor([1 if go else 0], dtype=torch.int32, device=dev)
            dist.broadcast(flag, 0)
            go = bool(flag.item())
    return n


def timed(fn, steps, world, dev):
    """EXACTLY `steps` calls of fn between barrier + synchronize pairs -> (elapsed seconds, max over ranks; last return value of fn)"""
    out = None
    _barrier(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    _barrier(world)
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, out


def instrument(L, n=3):
    """Per-kernel HIP-event timing cannot run inside graph replays: the same step is run eagerly (same kernels, same shapes, same stream)
    for `n` instrumented iterations -> (timer, n)."""
    from riders_amd import engine
    timer = engine.KernelTimer(repeat=L.args.timer_repeat)
    # one untimed eager step first: the eager path allocates tensors the graph's private pool never handed to the caching allocator, and a
    # hipMalloc inside an event pair was seen as a 0.9-ms "kernel" (configs[4] leg, 18 x the same launch's duration in four other runs)
    L.eager_step()
    torch.cuda.synchronize()
    engine.set_kernel_timer(timer)
    try:
        for _ in range(n):
            L.eager_step()
        torch.cuda.synchronize()
    finally:
        engine.set_kernel_timer(None)
    return timer, n


def measure_leg(L, steps, warmup, settle_seconds, world, rank, dev, prof_children=False):
    """warm-up, settle, timed region, instrumented steps, rooflines of one leg -> dict"""
    from riders_amd import engine
    args = L.args
    loss = None
    for _ in range(warmup):
        loss = L.step()
    nset = settle(L.step, settle_seconds, world, dev)
    timer = None
    if args.eager and not args.step_markers:
        timer = engine.KernelTimer(repeat=1)
        engine.set_kernel_timer(timer)
    if args.step_markers:      # profiler child: every step bracketed by the marker kernel (kernel_trace_child counts the dispatches in between)
        def marked():
            torch.cuda._sleep(1)
            return L.step()
        elapsed, loss = timed(marked, steps, world, dev)
        torch.cuda._sleep(1)
        torch.cuda.synchronize()
    else:
        elapsed, loss = timed(L.step, steps, world, dev)
    engine.set_kernel_timer(None)
    final_loss = float(loss.detach()) if loss is not None else float('nan')
    engine.check_roi_overflow()      # raises if a timed step met a RoI geometry the one-byte arg-max cannot encode (its gradients would be NaN)
    ms = elapsed * 1e3 / max(steps, 1)
    out = dict(value=L.batch_n * world * steps / elapsed, ms_per_step=ms, steps=steps, warmup=warmup, settle_steps=nset, final_loss=final_loss,
               batch_per_gpu=L.batch_n, height=L.h, width=L.w, launch_mode=L.launch_mode)
    tsteps = steps
    if timer is None and not args.step_markers:
        # EVERY rank runs the instrumented eager steps: with a reducer they contain collectives (a rank-0-only pass would wait for its peers forever)
        timer, tsteps = instrument(L)
    if rank == 0 and not args.step_markers:
        key = "%s_b%d_%dx%d_%s" % (L.kind, L.batch_n, L.h, L.w, args.dtype)
        prof = live = None
        if prof_children:
            tr = kernel_trace_child(args, L.kind, save_stats=args.save_kstats and os.path.join(args.save_kstats, "%s_kernel_stats.csv" % key))
            if tr is not None:
                prof = tr["per_kernel"]
                out["launches_per_step"], out["kernel_ms_per_step"], out["traced_wall_ms_per_step"] = tr["launches_per_step"], tr["kernel_ms_per_step"], tr["wall_ms_per_step"]
            if not args.no_live_traffic:
                live = live_traffic(args, L.kind)
        out.update(build_rooflines(timer, tsteps, ms, args.dtype, prof, live, key))
        if args.detail:
            rows = sorted(timer.detail().items(), key=lambda kv: -kv[1][1])
            suffix = "" if (L.kind == "rcnet" and not L.override) else "." + "_".join([L.kind] + [str(v) for k, v in sorted(L.override.items()) if k != "settle_seconds"])
            with open(args.detail + suffix, "w") as f:
                for (k, desc), (n, tms, fl, by) in rows:
                    f.write("%-12s %-48s launches/step %5.1f  ms/step %8.3f  TFLOP/s %7.2f  GB/s(alg) %8.1f\n" % (
                        k, desc, n / tsteps, tms / tsteps, fl / (tms * 1e-3) / 1e12 if tms > 0 else 0.0, by / (tms * 1e-3) / 1e9 if tms > 0 else 0.0))
    return out


def run_workload(kind, args, dev, world, rank, steps, warmup, prof_children=False, **override):
    """one stage on its own: build, measure, tear down"""
    from riders_amd import engine
    engine.clear_caches()
    L = build_leg(kind, args, dev, world, rank, **override)
    try:
        return measure_leg(L, steps, warmup, L.args.settle_seconds, world, rank, dev, prof_children)
    finally:
        close_leg(L)
        engine.set_param_grad_allocator(None)


def run_chain(args, dev, world, rank, steps, warmup):
    """BASELINE.json's metric: both stages alive in one process, ONE timed loop of (2 RC-Net steps + 1 SML step) = 16 images through both
    stages per iteration; then each stage timed on its own (same models, same graphs).  -> (chain, rcnet, sml) dicts"""
    from riders_amd import engine
    engine.clear_caches()
    rc = build_leg("rcnet", args, dev, world, rank)
    sm = build_leg("sml", args, dev, world, rank)
    try:
        def it():
            for _ in range(CHAIN_RC_STEPS):
                rc.step()
            return sm.step()
        for _ in range(warmup):
            it()
        nset = settle(it, args.settle_seconds, world, dev)
        elapsed, loss = timed(it, steps, world, dev)
        per = CHAIN_RC_STEPS * rc.batch_n
        if per != sm.batch_n:
            sys.stderr.write("bench.py: chain: %d RC-Net images vs %d SML images per iteration -- value counts the smaller\n" % (per, sm.batch_n))
        per = min(per, sm.batch_n)
        engine.check_roi_overflow()
        chain = dict(value=per * world * steps / elapsed, ms_per_step=elapsed * 1e3 / max(steps, 1), steps=steps, warmup=warmup, settle_steps=nset,
                     final_loss=float(loss.detach()), images_per_step=per, batch_per_gpu=per, height=rc.h, width=rc.w,
                     launch_mode="per stage: " + rc.launch_mode)
        r_rc = measure_leg(rc, steps, 0, min(args.settle_seconds, 1.0), world, rank, dev, prof_children=(world == 1 and not args.no_children))
        sm.args = argparse.Namespace(**dict(vars(sm.args), no_live_traffic=True))      # (kernel trace only for the SML stage: its counter passes would add ~1 min)
        r_sm = measure_leg(sm, steps, 0, min(args.settle_seconds, 1.0), world, rank, dev, prof_children=(world == 1 and not args.no_children))
        return chain, r_rc, r_sm
    finally:
        close_leg(rc); close_leg(sm)
        engine.set_param_grad_allocator(None)


def run_unchanged_caller(kind, args, dev, steps, warmup=3, autograph=False):
    """The path the reference's unmodified training loop takes through the aliased modules: per step the label build /
    device pre-step, model.forward, compute_loss, optimizer.zero_grad(), loss.backward() through torch.autograd (the engine's region is ONE
    autograd node), torch.optim.Adam.step() and loss.item() -- eager launches, no flat arena, no captured graph
    (RCNet/rcnet_main.py:342-359, train_zju.py:353-392).  autograph: with engine.set_autograph(True) -- one more line in INTEGRATION.md's aliasing
    block, none in the script -- the model's forward region is captured on its second call and replayed from then on (forward and backward
    hipGraphs inside the SAME torch.autograd / torch.optim.Adam loop).  -> dict(value, ms_per_step, ...)"""
    from riders_amd import engine, rcnet_main, sml_main
    engine.set_compute_dtype(args.dtype)
    engine.clear_caches()
    engine.set_param_grad_allocator(None)
    engine.set_autograph(bool(autograph))
    torch.manual_seed(0)
    if kind == "sml":
        import contextlib
        cfg = sml_main.ZJU_SML_CONFIG
        with contextlib.redirect_stdout(sys.stderr):
            model = sml_main.build_model(dev, cfg)
        batch = sml_main.synthetic_batch(args.sml_batch, args.sml_height, args.sml_width, seed=1234, device=dev)
        outlier = sml_main.make_outlier_removal(cfg)
        fwd = lambda: sml_main.forward_loss(model, batch, cfg, outlier)      # noqa: E731
        n, h, w = args.sml_batch, args.sml_height, args.sml_width
    else:
        cfg = rcnet_main.ZJU_CONFIG
        model = rcnet_main.build_model(dev, cfg)
        batch = rcnet_main.synthetic_batch(args.batch, args.height, args.width, cfg, seed=1234, device=dev)
        fwd = lambda: rcnet_main.forward_loss(model, batch, cfg)      # noqa: E731
        n, h, w = args.batch, args.height, args.width
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=cfg['learning_rate'])
    last = [0.0]

    def step():
        loss = fwd()
        opt.zero_grad()
        loss.backward()
        opt.step()
        last[0] = loss.item()      # the reference logs the loss every step: one host synchronisation per step
        return loss
    try:
        for _ in range(warmup):
            step()
        if args.step_markers:
            def marked():
                torch.cuda._sleep(1)
                return step()
            elapsed, _ = timed(marked, steps, 1, dev)
            torch.cuda._sleep(1)
            torch.cuda.synchronize()
        else:
            elapsed, _ = timed(step, steps, 1, dev)
        return dict(value=n * steps / elapsed, ms_per_step=elapsed * 1e3 / steps, steps=steps, warmup=warmup, settle_steps=0, final_loss=last[0], batch_per_gpu=n,
                    height=h, width=w, autograph=engine.autograph_stats() if autograph else None,
                    launch_mode=("torch.autograd + torch.optim.Adam + loss.item() per step; the model's forward region captured and replayed (engine.set_autograph)"
                                 if autograph else "eager through torch.autograd, torch.optim.Adam, loss.item() per step"))
    finally:
        del opt, model
        engine.set_autograph(False)
        engine.clear_caches()
        torch.cuda.empty_cache()


# ================================================================================================ launching
def visible_gpu_count():
    """GPUs this process may use, WITHOUT touching the HIP / HSA runtime (torch.cuda.device_count() falls back to hipGetDeviceCount on ROCm
    builds without amdsmi, which initialises the runtime in the parent): the kfd topology lists one node per agent, GPUs are the nodes with
    SIMDs; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow it.  None when the topology cannot be read."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for p in nodes:
        try:
            props = dict(l.split()[:2] for l in open(p).read().splitlines() if len(l.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(n, argv):
    """GPU-free parent: start n rank processes (one per GPU) BEFORE anything in this process touches a GPU, relay rank 0's stdout; when a
    rank exits with an error the others are terminated at once (they would sit in RCCL until its timeout) and that code is returned."""
    import socket
    import subprocess
    have = visible_gpu_count()
    if have is None:
        have = torch.cuda.device_count()      # last resort (may initialise the runtime; starting fresh child processes afterwards is still fine)
    if have < n and "--share-gpu" not in argv:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible\n" % (n, have))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL across processes)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst = 0
    while procs:
        time.sleep(0.2)
        for p in list(procs):
            rc = p.poll()
            if rc is None:
                continue
            procs.remove(p)
            if rc != 0:
                worst = max(worst, abs(rc))
                for q_ in procs:
                    q_.terminate()
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-seconds", type=float, default=2.0, help="untimed replays before the timed region (clock settling)")
    ap.add_argument("--batch", type=int, default=8, help="RC-Net images per GPU and step (BASELINE configs[1]: 8)")
    ap.add_argument("--config3", action="store_true", help="BASELINE configs[3]: global batch 32 over the ranks (4 images per rank on 8 GPUs)")
    ap.add_argument("--dtype", default=os.environ.get("RIDERS_BENCH_DTYPE", "bf16"), choices=["fp32", "bf16", "fp16"],
                    help="activation dtype (BASELINE.json configs[1] quotes bf16; fp32 is the 1e-3 parity mode; fp16 = configs[4], with --height 512 "
                         "--width 1024, static loss scale --loss-scale)")
    ap.add_argument("--loss-scale", type=float, default=None, help="static loss scale (default 16384 for fp16, 1 otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--workload", default="chain", choices=["chain", "rcnet", "sml", "caller_rcnet", "caller_sml"],
                    help="chain (default) = BASELINE.json's metric: RC-Net steps + SML step in one timed loop; rcnet = configs[1] alone; sml = configs[2] alone; "
                         "caller_* = the unchanged caller's eager path (torch.autograd + torch.optim.Adam + loss.item())")
    ap.add_argument("--no-sml", action="store_true", help="(kept for tools) with --workload rcnet nothing else is run anyway")
    ap.add_argument("--no-legs", action="store_true", help="skip the fp32 (parity mode), configs[4] (fp16 512x1024) and unchanged-caller entries of the default N = 1 run")
    ap.add_argument("--sml-batch", type=int, default=16)
    ap.add_argument("--sml-height", type=int, default=256)
    ap.add_argument("--sml-width", type=int, default=512)
    ap.add_argument("--eager", action="store_true", help="do not capture forward+backward into hipGraphs")
    ap.add_argument("--force-ddp", action="store_true", help="run the N > 1 code path (RCCL process group, stage-bucketed all-reduce) on however "
                                                              "many ranks there are, including one: a functional check of that path on a 1-GPU box")
    ap.add_argument("--allreduce", default="all_reduce", choices=["all_reduce", "rs_ag"],
                    help="gradient exchange per bucket: one all-reduce (RCCL's choice: a ring on xGMI) or reduce_scatter + all_gather in place")
    ap.add_argument("--full-json", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="where the complete record (all tables) is written; stdout carries the compact line only")
    ap.add_argument("--opts", default=os.environ.get("RIDERS_OPTS", ""),
                    help="A/B switches, 'name=value,...': engine switches (engine.set_switch: lazy_bn, roi_u8, ...) and, prefixed rd., routing options of "
                         "the library (rd_set_option: rd.frag_v128, rd.frag32_v128, ...); validated, unknown names fail.  Default: $RIDERS_OPTS")
    ap.add_argument("--comm", default="c_abi", choices=["c_abi", "torch"],
                    help="gradient exchange transport for N > 1: c_abi = the library's own RCCL communicator (rd_comm_*, collectives captured into the "
                         "step's ONE hipGraph), torch = torch.distributed's nccl(=RCCL) collectives between per-stage graphs (round 4)")
    ap.add_argument("--pg-backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend; gloo (with --share-gpu) lets the N > 1 CONTROL FLOW -- "
                    "chain loop, settle broadcast, barriers, bucketed reducers, max over ranks -- run on a one-GPU box (collectives through the host; forces --comm torch)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (control-flow check of the N > 1 path on a one-GPU box; never a measurement)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not start the rocprofv3 counter passes (FETCH_SIZE / WRITE_SIZE / MFMA-busy children) that "
                    "measure roofline.traffic inside the default N = 1 run")
    ap.add_argument("--no-children", action="store_true", help="start no profiler child at all (kernel trace, counter passes): durations then come from HIP events "
                    "only.  Pass it in every command that is itself run under rocprofv3 (it is also detected)")
    ap.add_argument("--autograph", action="store_true", help="(caller_* workloads) engine.set_autograph(True): the forward region is captured and replayed inside the unchanged loop")
    ap.add_argument("--step-markers", action="store_true", help="(profiler children) bracket every timed step with torch.cuda._sleep(1) so that a kernel trace can be cut per step")
    ap.add_argument("--save-kstats", default=None, help="directory that receives the kernel-trace child's *_kernel_stats.csv (the summary committed under profiles/)")
    ap.add_argument("--detail", default=None, help="write a per-launch-shape timing table to this file")
    ap.add_argument("--timer-repeat", type=int, default=5, help="idempotent launches issued this many times per HIP-event pair in the instrumented "
                                                                "steps (1 under rocprofv3, so that its launch counts per step are the real ones)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE line: whatever a library writes to file descriptor 1 during the run (RCCL's five-line version banner at the first
    # communicator) goes to stderr; the descriptor is restored right in front of the final print
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` (it spawns its ranks) or under "
                         "torch.distributed.run with --nproc-per-node equal to --gpus\n" % (args.gpus, world))
        sys.exit(2)
    if args.config3:
        if 32 % world:
            sys.stderr.write("bench.py: --config3 needs a world size dividing 32\n")
            sys.exit(2)
        args.batch = 32 // world
        args.sml_batch = CHAIN_RC_STEPS * args.batch
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the riders_amd hot path has no CPU fallback")
    dev = torch.device("cuda", 0 if args.share_gpu else local_rank)
    torch.cuda.set_device(dev)
    if args.pg_backend == "gloo":
        args.comm = "torch"
    if under_profiler():
        args.no_children = True
    ddp = world > 1 or args.force_ddp
    comm = None
    if ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.pg_backend == "gloo":
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        assert dist.get_world_size() == world
        comm = dict(backend="nccl (RCCL)" if args.pg_backend == "nccl" else "gloo (control-flow check, not a measurement)", world_size=dist.get_world_size(),
                    rccl_version=".".join(str(v) for v in torch.cuda.nccl.version()))
    args.rccl_comm = None
    if ddp and args.comm == "c_abi":
        from riders_amd.parallel import RcclComm
        try:      # RcclComm fails on EVERY rank together (the ranks vote before rd_comm_init), so the fall-back below is the same decision everywhere
            args.rccl_comm = RcclComm(rank, world)
        except Exception as ex:
            sys.stderr.write("bench.py: rank %d: no C-ABI communicator (%r); falling back to --comm torch\n" % (rank, ex))
        comm["transport"] = "rd_comm (C ABI, captured)" if args.rccl_comm is not None else "torch.distributed"
    if ddp:
        # every rank must see the world it was asked for before anything is timed
        assert comm["world_size"] == args.gpus, "communicator world size %d != --gpus %d" % (comm["world_size"], args.gpus)

    if args.opts:
        from riders_amd import engine
        engine.apply_opts(args.opts)
    rc = sml = None
    legs, caller = {}, None
    if args.workload == "chain":
        head, rc, sml = run_chain(args, dev, world, rank, args.steps, args.warmup)
    elif args.workload.startswith("caller"):
        head = run_unchanged_caller("sml" if args.workload.endswith("sml") else "rcnet", args, dev, args.steps, min(args.warmup, 3), autograph=args.autograph)
    else:
        head = run_workload(args.workload, args, dev, world, rank, args.steps, args.warmup, prof_children=(world == 1 and not args.no_children and not args.eager))
    default_line = args.workload == "chain" and world == 1 and args.dtype == "bf16" and (args.height, args.width, args.batch) == (256, 512, 8) \
        and not args.config3 and not args.force_ddp and not args.eager
    if world > 1:      # every rank ran the same number of settle steps (and therefore of collectives): checked, not assumed
        import torch.distributed as dist
        mine = torch.tensor([int(head.get("settle_steps") or 0)], dtype=torch.int64, device=dev)
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert int(lo) == int(hi), "ranks ran different numbers of settle steps (%d .. %d)" % (int(lo), int(hi))
    if default_line and not args.no_legs:
        # every mode that carries a claim is timed by the same run: the fp32 parity mode (north_star's 1e-3 holds for it) and BASELINE
        # configs[4] (fp16, 512x1024, batch 8 per GPU); shorter legs, same timing method
        sec_steps = max(1, min(args.steps, 60))
        legs["fp32"] = run_workload("rcnet", args, dev, world, rank, sec_steps, min(args.warmup, 5), dtype="fp32", settle_seconds=min(args.settle_seconds, 1.0))
        legs["config4"] = run_workload("rcnet", args, dev, world, rank, sec_steps, min(args.warmup, 5), dtype="fp16", height=512, width=1024,
                                       settle_seconds=min(args.settle_seconds, 1.0))
        # ... and its SML half at the same per-rank size (8 frames of 512x1024, fp16, static loss scale 1024)
        legs["config4_sml"] = run_workload("sml", args, dev, world, rank, min(sec_steps, 30), min(args.warmup, 3), dtype="fp16", sml_batch=8, sml_height=512,
                                           sml_width=1024, sml_native=True, loss_scale=1024.0, settle_seconds=min(args.settle_seconds, 1.0))
        # the path an UNCHANGED training script takes (eager, torch.optim.Adam, loss.item() per step) next to the graphed step
        caller = {}
        for kind, graphed in (("rcnet", rc), ("sml", sml)):
            try:
                nst = max(5, min(args.steps, 30))
                c = run_unchanged_caller(kind, args, dev, nst, autograph=True)      # the aliasing block's setting: regions captured on their second call
                ce = run_unchanged_caller(kind, args, dev, nst, autograph=False)    # plain eager launches
                c["eager_value"], c["eager_ms_per_step"] = ce["value"], ce["ms_per_step"]
                c["graphed_value"] = graphed["value"]
                if not args.no_children:
                    for tag, ag in (("", True), ("eager_", False)):
                        tr = kernel_trace_child(argparse.Namespace(**dict(vars(args), autograph=ag)), "caller_" + kind, steps=8)
                        if tr is not None:
                            c[tag + "launches_per_step"], c[tag + "kernel_ms_per_step"] = tr["launches_per_step"], tr["kernel_ms_per_step"]
                caller[kind] = c
            except Exception as ex:      # never lose the bench line to an auxiliary leg
                caller[kind] = dict(error=repr(ex)[:300])
    if rank == 0:
        cpu = val = None
        if world == 1 and not args.no_cpu_baseline and args.workload in ("chain", "rcnet"):
            try:
                val = val_abs_rel_pair(dev)
            except Exception as ex:      # never lose the bench line to the auxiliary figure
                val = dict(error=repr(ex)[:300])
            cpu = cpu_baseline()
        full = full_record(args, world, comm, ddp, head, rc, sml, legs, cpu, val, caller)
        line = compact_line(full)
        # the complete record (per-kernel / per-family / per-shape tables, notes, PMC blocks) goes to the --full-json file; stdout carries
        # ONE line the driver can hold (round 4's 39.8-KB line was not parsed: BENCH_r04.json.parsed = null)
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.full_json)), exist_ok=True)
            with open(args.full_json, "w") as f:
                json.dump(full, f)
        except OSError as ex:
            sys.stderr.write("bench.py: could not write %s: %r\n" % (args.full_json, ex))
        sys.stderr.write("bench.py: full record (%d bytes) -> %s\n" % (len(json.dumps(full)), args.full_json))
        sys.stderr.flush()
        try:      # C-level stdout first (RCCL prints its version banner through stdio: buffered, it would otherwise land behind the line at exit)
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(render_line(line), flush=True)
        os.dup2(2, 1)      # (communicator teardown may print again)
    if ddp:
        import torch.distributed as dist
        if args.rccl_comm is not None:
            try:
                torch.cuda.synchronize()
                args.rccl_comm.close()
            except Exception as ex:
                sys.stderr.write("bench.py: rd_comm_destroy: %r\n" % (ex,))
        # the line is out; quiesce before the group goes away and never let a teardown problem turn into the run's exit code
        try:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            dist.destroy_process_group()
        except Exception as ex:
            sys.stderr.write("bench.py: process-group teardown: %r\n" % (ex,))


if __name__ == "__main__":
    main()
