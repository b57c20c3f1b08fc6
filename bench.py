"""bench.py -- RC-Net (+ Scale Map Learner) training throughput on MI355X (BASELINE.json metric: train imgs/sec, 256x512).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full optimisation step of the hot path on one batch of synthetic input already resident in HBM:
/255 normalise + label build + RC-Net forward + masked BCE + backward + (RCCL gradient all-reduce for N > 1, started
per stage while the backward is still running) + fused Adam.
Workload at N = 1 is BASELINE.json configs[1]: RC-Net training, batch 8 per GPU (K = 30 radar points, patch 240x100,
3x256x512 thermal image edge-padded to 496x612), bf16.  Weak scaling: every rank processes its own batch of 8
(configs[3] quotes a global batch of 32 on 8 GPUs = 4 per rank; `--batch 4` runs that; the default keeps the per-rank
work of the N = 1 line so that per-N values are comparable).

Timing: W warm-up steps, then untimed "settling" replays until `--settle-seconds` of GPU work have passed (clocks and the
SMI sampler settle; a 20-step region is only 0.2 s), then EXACTLY K timed steps between barrier + synchronize pairs.

Launching: with WORLD_SIZE in the environment (torch.distributed.run) this process is one rank and `--gpus` must equal WORLD_SIZE (exit 2
otherwise).  Without it, `--gpus N` (N > 1) makes THIS process a GPU-free parent that starts N rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment) before anything touches a GPU, relays rank 0's JSON line and exits with the
worst rank's code.

Prints ONE JSON line on rank 0 (see the contract in the task statement) including
  "roofline":     the DOMINANT KERNEL of the step = the kernel instantiation (name as in a rocprofv3 kernel trace, from
                  rd_conv_fwd_kernel_name / rd_conv_wgrad_kernel_name) with the largest summed duration: launches per step, average launch
                  duration (HIP events on the launch stream; idempotent launches are issued 5x between one event pair so the few
                  microseconds of the event pair do not inflate 20-100 us kernels), algorithmic FLOPs (2 / MAC) and bytes (each operand
                  once) of exactly those launches, and the fraction of the roof that binds them (HBM 8 TB/s below 312 FLOP/B, dense MFMA
                  2.5 PFLOP/s above).  `families` keeps the per-family table, `mfma_busy` the PMC figure of the committed rocprofv3 pass.
  "cpu_baseline": the oracle (PyTorch-CPU restatement of the reference path) timed on this box's host cores on a bounded sample: best
                  over thread counts {8, 16, 32, 64, all} for RC-Net B = 1, then B = 8 and an SML step at the best count (test
                  infrastructure used as the reported baseline only), plus the validation chain's abs-rel on both paths.
  "sml":          the same measurement for BASELINE.json configs[2] (Scale Map Learner, batch 16, bf16) at 256x512 (N = 1 only).
  "chained":      images/s through both stages (RC-Net step + SML step per image), the figure BASELINE.json's metric names.
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
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r05_traffic.json")   # written by tools/traffic_pass.sh (rocprofv3 --pmc passes)
PMC_FILE = os.path.join(ROOT, "profiles", "r05_pmc_dominant.json")   # MFMA-busy / wait counters of the dominant kernel (tools/pmc_dominant.sh)
KSTATS_FILE = "profiles/r05_%s_kernel_stats.csv"                     # rocprofv3 --kernel-trace --stats summary of the same command
RIDGE = {k: v * 1e12 / (PEAK_HBM * 1e9) for k, v in PEAK_MFMA.items()}     # FLOP/B above which the MFMA roof binds


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


def val_abs_rel_pair(dev):
    """Validation chain (val_zju.py:124-254: device pre-step -> network (eval) -> 1/pred -> bicubic -> masked metrics) on identical random-init
    weights and synthetic frames: abs-rel of the HIP path (fp32 parity mode) and of the oracle chain.  north_star: within 1e-3."""
    import contextlib
    from oracle import sml as OS
    from riders_amd import engine, sml_main
    engine.set_compute_dtype("fp32"); engine.clear_caches()
    torch.manual_seed(5)
    with contextlib.redirect_stdout(sys.stderr):
        m = sml_main.build_model(dev, sml_main.ZJU_SML_CONFIG)
    m.eval()
    B, H, W = 2, 60, 80
    batch = sml_main.synthetic_batch(B, H, W, seed=21)
    got = sml_main.validate_batch(m, tuple(b.to(dev) for b in batch))
    o = OS.SMLOracle()
    o.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    o.eval()
    image, mono, radar, gt, sparse_gt, rcnet = [b.numpy() for b in batch]
    hw = sml_main.net_size(H, W)
    hip, ref = [], []
    for i in range(B):
        xo, do, _ = OS.prestep_sample(image[i], mono[i, 0], radar[i, 0], rcnet[i, 0], hw)
        with torch.no_grad():
            po = o(torch.from_numpy(xo)[None].float(), torch.from_numpy(do)[None].float())
        ref.append(float(OS.val_metrics(po, sparse_gt[i, 0], (H, W))["abs_rel"]))
        hip.append(float(got["abs_rel"][i]))
    engine.clear_caches()
    return dict(hip=sum(hip) / B, oracle=sum(ref) / B, max_abs_diff=max(abs(a - b) for a, b in zip(hip, ref)),
                sample="%d synthetic %dx%d frames, random-init weights (identical on both paths), fp32" % (B, H, W))


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
        except Exception as ex:      # the baseline leg never fails the bench line
            out["sml_b1"] = dict(error=repr(ex)[:200])
    torch.set_num_threads(keep)
    out["seconds"] = time.time() - t_start
    return out


def family_table(timer, timed_steps, ms_per_step, dtype):
    """Per-family table from the HIP-event records (family = the `kind` the engine tags a launch with)."""
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


def kernel_roofline(timer, timed_steps, ms_per_step, dtype, traffic_key, conv_only=False, with_tables=True):
    """The roofline object: the named kernel with the largest summed launch time (see the module docstring).  conv_only: among the
    convolution / weight-gradient kernels only (`roofline_conv`); otherwise among ALL named kernels incl. the BatchNorm passes."""
    peak_mfma = PEAK_MFMA[dtype]
    ks = timer.by_kernel()
    if conv_only:
        ks = {n: v for n, v in ks.items() if v.get("kind") in ("conv_gemm", "conv_wgrad")}
    fams = family_table(timer, timed_steps, ms_per_step, dtype) if with_tables else None
    if not ks:
        return dict(families=fams) if fams else None
    dom = max(ks, key=lambda k: ks[k]["ms"])
    k = ks[dom]
    intensity = k["flops"] / max(k["bytes"], 1.0)
    bound = "mfma" if intensity >= RIDGE[dtype] else "hbm"
    ach = k["flops"] / (k["ms"] * 1e-3) / 1e12 if bound == "mfma" else k["bytes"] / (k["ms"] * 1e-3) / 1e9
    peak = peak_mfma if bound == "mfma" else PEAK_HBM
    top = sorted(k["shapes"].items(), key=lambda kv: -kv[1][1])[:4]
    roof = dict(bound=bound, kernel=dom, achieved=ach, peak=peak, unit="TFLOP/s" if bound == "mfma" else "GB/s", frac=ach / peak, traffic=None,
                launches_per_step=k["launches"] / timed_steps, avg_launch_us=k["ms"] * 1e3 / k["launches"], ms_per_step=k["ms"] / timed_steps,
                share_of_step=(k["ms"] / timed_steps) / ms_per_step,
                algorithmic_flops_per_launch=k["flops"] / k["launches"], algorithmic_bytes_per_launch=k["bytes"] / k["launches"],
                flops_per_byte=intensity,
                shapes=[dict(shape=d, launches_per_step=v[0] / timed_steps, avg_us=v[1] * 1e3 / v[0], tflops=v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0,
                             gbs=v[3] / (v[1] * 1e-3) / 1e9 if v[1] > 0 else 0.0) for d, v in top],
                kernels={n: dict(ms_per_step=v["ms"] / timed_steps, launches_per_step=v["launches"] / timed_steps, avg_launch_us=v["ms"] * 1e3 / v["launches"],
                                 tflops=v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0,
                                 gbs=v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0,
                                 frac=max(v["flops"] / (v["ms"] * 1e-3) / 1e12 / peak_mfma, v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM) if v["ms"] > 0 else 0.0)
                         for n, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms"])[:10]} if with_tables else None,
                families=fams,
                note="dominant = the %s kernel instantiation with the largest summed launch time over %d instrumented eager "
                     "steps of the same workload right after the timed region; durations from HIP events on the launch stream, idempotent launches "
                     "issued %dx per event pair (the BatchNorm backward's reduce / finalize / apply launches are issued and timed one by one); "
                     "achieved = algorithmic FLOPs (2/MAC) or bytes (each operand once) of exactly those launches / "
                     "their summed durations; the rocprofv3 --kernel-trace --stats summary of the same command is %s"
                     % ("convolution / weight-gradient" if conv_only else "named (convolution, weight-gradient, BatchNorm-pass)", timed_steps, timer.repeat,
                        KSTATS_FILE % traffic_key))
    if not with_tables:
        roof.pop("kernels", None); roof.pop("families", None)
    try:    # PMC figures of the committed rocprofv3 counter passes (they cannot be collected inside this run)
        pm = json.load(open(PMC_FILE)).get(traffic_key.split("_")[0], {}).get(dom)
        if pm:
            roof["mfma_busy"] = pm
    except (OSError, ValueError):
        pass
    try:    # HBM bytes per launch from PMC counters: separate rocprofv3 passes of the same command (tools/traffic_pass.sh), per family
        tj = json.load(open(TRAFFIC_FILE)).get(traffic_key, {})
        tr = tj.get("kernels", {}).get(dom) or None
        if tr is not None:
            roof["traffic"] = tr["bytes_per_launch"]
            roof["traffic_over_algorithmic"] = tr["bytes_per_launch"] / max(roof["algorithmic_bytes_per_launch"], 1.0)
            roof["traffic_source"] = tj.get("note", "")
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from traffic_aggregate import csrc_fingerprint
            roof["traffic_stale"] = tj.get("_csrc_sha1") != csrc_fingerprint(ROOT)
    except (OSError, ValueError, KeyError):
        pass
    return roof


_PROFILER_VARS = ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY", "ROCPROF_", "ROCPROFILER_", "ROCP_", "HSA_TOOLS_LIB", "ROCTX_")


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


def live_traffic(args, kernels, budget_s=45.0):
    """HBM bytes per launch and matrix-pipe busy fraction of the named kernels, measured by THIS run: three rocprofv3 counter passes (FETCH_SIZE;
    WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE -- separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes) of the same
    workload as child processes (3 eager steps each), parsed with the units / gfx950 correction of tools/traffic_aggregate.py and the
    normalisation of tools/pmc_dominant.py.  -> {kernel: dict(bytes_per_launch, dispatches[, mfma_util])} or None (no rocprofv3, a child failed
    or ran out of its time budget: the committed profile values stay).  Children only -- this process holds the GPU and never replaces itself."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    if under_profiler():      # this run is itself being profiled: a counter pass started from here would be a profiler inside a profiler
        sys.stderr.write("bench.py: running under a profiler -- the live counter passes are skipped\n")
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from traffic_aggregate import kernel_key
    out = tempfile.mkdtemp(prefix="riders_live_traffic_", dir="/tmp")
    base = ["python3", os.path.abspath(__file__), "--workload", "rcnet", "--eager", "--steps", "2", "--warmup", "1", "--settle-seconds", "0", "--no-cpu-baseline",
            "--no-sml", "--no-legs", "--no-live-traffic", "--batch", str(args.batch), "--dtype", args.dtype, "--height", str(args.height), "--width", str(args.width),
            "--full-json", os.path.join(out, "child_full.json")]
    if args.opts:
        base += ["--opts", args.opts]
    per = {}
    try:
        for n, ctrs in enumerate((("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))):
            d = os.path.join(out, "pass%d" % n)
            cmd = ["rocprofv3", "--pmc"] + list(ctrs) + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + base
            pr = subprocess.Popen(cmd, cwd="/tmp", env=dict(clean_profiler_env(os.environ), TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                  start_new_session=True)
            try:
                rc = pr.wait(timeout=budget_s)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)
                pr.wait()
                rc = -1
            found = False
            if rc != 0 and n < 2:      # a failed / timed-out traffic pass: do not spend another budget on the next one
                return None
            if rc == 0:
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f, newline="")):
                        if row["Counter_Name"] not in ctrs or "rd" not in row["Kernel_Name"]:
                            continue
                        k = kernel_key(row["Kernel_Name"])
                        if k in kernels:
                            e = per.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
                            e[0] += float(row["Counter_Value"]); e[1] += 1
                            found = True
            if not found and n < 2:      # the traffic passes are the point; the matrix-pipe pass is an extra
                return None
    except Exception as ex:      # never lose the bench line to an auxiliary measurement
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
                res[k]["mfma_util"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
    return res or None


def apply_live_traffic(roof, live):
    """overwrite a roofline object's `traffic` (committed counter file) with this run's own counter passes"""
    if not roof or not live or roof.get("kernel") not in live:
        return
    t = live[roof["kernel"]]
    if roof.get("traffic") is not None:
        roof["traffic_committed"] = roof["traffic"]
    roof["traffic"] = t["bytes_per_launch"]
    roof["traffic_over_algorithmic"] = t["bytes_per_launch"] / max(roof.get("algorithmic_bytes_per_launch", 0.0), 1.0)
    roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (--kernel-trace only) of the same workload, started as child processes by THIS "
                              "bench run after its timed regions: mean over %d dispatches; FETCH_SIZE doubled (gfx950)" % t["dispatches"])
    roof["traffic_stale"] = False
    roof["traffic_live"] = True
    if "mfma_util" in t:
        mb = dict(roof.get("mfma_busy") or {})
        if "mfma_util" in mb:
            mb["mfma_util_committed"] = mb["mfma_util"]
        mb["mfma_util"] = t["mfma_util"]
        mb["mfma_util_source"] = "live: SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8) of this run's own counter pass"
        roof["mfma_busy"] = mb


LINE_LIMIT = 6000      # bytes of the ONE stdout line (the driver kept ~8 KB of stdout and lost the head of a 39.8-KB line)
FLAT_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "traffic_stale", "launches_per_step",
                  "avg_launch_us", "ms_per_step", "share_of_step", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch", "mfma_util", "traffic_live")


def _sig(v, n=5):
    """floats to n significant digits (the line is for a parser with a size limit, not for arithmetic)"""
    if isinstance(v, float):
        return float("%.*g" % (n, v)) if v == v and abs(v) != float("inf") else None
    return v


def flat_roofline(roof, keys=FLAT_ROOF_KEYS):
    """the flat part of a roofline object: no kernels / families / shapes tables, no notes; `mfma_util` lifted out of the PMC block"""
    if not roof or "kernel" not in roof:
        return None
    r = dict(roof)
    if isinstance(r.get("mfma_busy"), dict):
        r["mfma_util"] = r["mfma_busy"].get("mfma_util")
    return {k: _sig(r[k]) for k in keys if k in r}


def full_record(args, world, comm, ddp, head, sml, legs, cpu, val):
    """Everything the run measured (what round 4 printed on stdout): written to the --full-json file."""
    is_rc = args.workload == "rcnet"
    out = {
        "metric": "train imgs/sec (RC-Net, 256x512 thermal + 30 radar points, patch 240x100)" if is_rc else
                  "train imgs/sec (Scale Map Learner, MiDaS-small / EfficientNet-Lite3)",
        "value": head["value"], "unit": "imgs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16": "bf16", "fp16": "f16"}[args.dtype], "data": "synthetic",
        "config": {"workload": ("RC-Net training step, batch %d/GPU, ZJU config (K=30, patch 240x100), %dx%d image, fwd+loss+bwd+Adam" if is_rc
                                else "SML training step, batch %d/GPU, %dx%d frames, device pre-step+fwd+loss+bwd+Adam") % (
            head["batch_per_gpu"], head["height"], head["width"]), "global_batch": head["batch_per_gpu"] * world,
            "parallelism": "dp%d" % world,
            "note": ("BASELINE configs[3]: global batch 32 over %d rank(s)" % world) if args.config3 else
                    "BASELINE configs[1] per rank at every N (weak scaling); configs[3]'s global 32 on 8 GPUs is --config3"},
        "final_loss": head["final_loss"], "launch_mode": head["launch_mode"], "settle_steps": head["settle_steps"],
        "world_size": world, "comm": comm,
        "allreduce": None if not ddp else "RCCL sum (%s, %s) of the flat fp32 gradient arena in 3 stage buckets, each started when the backward "
                                          "passes its stage mark (overlaps the remaining backward); 1/N folded into Adam" % (
                                              args.allreduce, "rd_allreduce_bucket captured in the step graph" if getattr(args, "rccl_comm", None) is not None
                                              else "torch.distributed between stage graphs"),
        "roofline": head.get("roofline"), "roofline_conv": head.get("roofline_conv"),
    }
    if sml is not None:
        out["sml"] = {"metric": "train imgs/sec (Scale Map Learner, MiDaS-small / EfficientNet-Lite3; BASELINE configs[2])",
                      "value": sml["value"], "unit": "imgs/s", "ms_per_step": sml["ms_per_step"], "dtype": out["dtype"],
                      "config": {"workload": "SML training step, batch %d/GPU, %dx%d frames, device pre-step+fwd+loss+bwd+Adam" % (
                          sml["batch_per_gpu"], sml["height"], sml["width"])},
                      "final_loss": sml["final_loss"], "settle_steps": sml["settle_steps"], "roofline": sml.get("roofline"),
                      "roofline_conv": sml.get("roofline_conv")}
        # BASELINE.json's metric names both stages: an image passes through an RC-Net step and an SML step
        out["chained"] = {"metric": "train imgs/sec through RC-Net then SML (per-image time = RC-Net step/8 + SML step/16)",
                          "value": 1.0 / (1.0 / head["value"] + 1.0 / sml["value"]), "unit": "imgs/s"}
    for name, leg in legs.items():
        out[name] = {"metric": LEG_METRIC[name], "value": leg["value"], "unit": "imgs/s", "ms_per_step": leg["ms_per_step"], "steps": leg["steps"],
                     "dtype": LEG_DTYPE[name],
                     "config": {"workload": "%s training step, batch %d/GPU, %dx%d image, fwd+loss+bwd+Adam" % (
                         "SML" if name.endswith("_sml") else "RC-Net", leg["batch_per_gpu"], leg["height"], leg["width"])},
                     "final_loss": leg["final_loss"], "roofline": leg.get("roofline"), "roofline_conv": leg.get("roofline_conv")}
    if val is not None:
        out["val_abs_rel"] = val
    if cpu is not None:
        out["cpu_baseline"] = cpu
    return out


LEG_METRIC = {"fp32": "train imgs/sec (RC-Net, batch 8, 256x512, fp32: the 1e-3 parity mode)",
              "config4": "train imgs/sec (RC-Net, batch 8 per GPU, 3x512x1024, fp16: BASELINE configs[4] per rank)",
              "config4_sml": "train imgs/sec (Scale Map Learner, batch 8 per GPU, 512x1024 at native resolution, fp16: BASELINE configs[4] per rank)"}
LEG_DTYPE = {"fp32": "f32", "config4": "f16", "config4_sml": "f16"}


def compact_line(full):
    """The ONE stdout line: headline fields, flat `roofline` / `roofline_conv`, `cpu_baseline` (value, unit, cores, kind, sample) and for the
    secondary legs only value / ms_per_step / dtype / config.workload / roofline.{kernel, bound, frac}.  Always < LINE_LIMIT bytes."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _sig(full[k], 7) for k in keep}
    cfg = full["config"]
    line["config"] = {"workload": cfg["workload"], "global_batch": cfg["global_batch"], "parallelism": cfg["parallelism"]}
    line["roofline"] = flat_roofline(full.get("roofline"))
    line["roofline_conv"] = flat_roofline(full.get("roofline_conv"))
    cb = full.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = {k: _sig(cb[k]) for k in ("value", "unit", "cores", "kind", "sample", "host_cpus") if k in cb}
    for name in ("sml", "fp32", "config4", "config4_sml"):
        leg = full.get(name)
        if leg is None:
            continue
        r = leg.get("roofline") or {}
        line[name] = {"value": _sig(leg["value"], 7), "unit": leg.get("unit", "imgs/s"), "ms_per_step": _sig(leg["ms_per_step"], 7), "dtype": leg["dtype"],
                      "config": {"workload": leg["config"]["workload"]},
                      "roofline": {k: _sig(r[k]) for k in ("kernel", "bound", "frac") if k in r}}
    if "chained" in full:
        line["chained"] = {"value": _sig(full["chained"]["value"], 7), "unit": "imgs/s"}
    va = full.get("val_abs_rel")
    if isinstance(va, dict) and "hip" in va:
        line["val_abs_rel"] = {k: _sig(va[k], 6) for k in ("hip", "oracle", "max_abs_diff")}
    for k in ("final_loss", "launch_mode", "world_size"):
        line[k] = _sig(full.get(k), 7)
    if full.get("comm"):
        line["comm"] = full["comm"]
    return line


def render_line(line, limit=LINE_LIMIT):
    """json text of the compact line; should a field ever grow past the limit, optional blocks are dropped (last first) rather than the line lost"""
    text = json.dumps(line, separators=(",", ":"))
    for k in ("comm", "launch_mode", "val_abs_rel", "chained", "config4_sml", "config4", "fp32", "sml", "roofline_conv"):
        if len(text) < limit:
            break
        line = {a: b for a, b in line.items() if a != k}
        text = json.dumps(line, separators=(",", ":"))
    return text



def run_workload(kind, args, dev, world, rank, steps, warmup, **override):
    """override: dtype / batch / height / width / settle_seconds of a secondary leg (the fp32 parity mode, configs[4]) on a copy of args."""
    from riders_amd import engine, rcnet_main, sml_main
    if override:
        args = argparse.Namespace(**dict(vars(args), **override))
    from riders_amd.optim import FlatAdam
    from riders_amd.parallel import GradientAllReducer, rcnet_stages, sml_stages
    engine.set_compute_dtype(args.dtype)
    engine.clear_caches()
    torch.manual_seed(0)  # identical initial weights on every rank
    if kind == "sml":
        batch_n, h, w = args.sml_batch, args.sml_height, args.sml_width
        cfg = dict(sml_main.ZJU_SML_CONFIG, net_hw=(h, w)) if getattr(args, "sml_native", False) else sml_main.ZJU_SML_CONFIG
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):      # the constructor prints like the reference's; stdout carries the ONE JSON line only
            model = sml_main.build_model(dev, cfg)
        main_mod, extra, stages = sml_main, dict(outlier=sml_main.make_outlier_removal(cfg)), sml_stages(model)
        batch = sml_main.synthetic_batch(batch_n, h, w, seed=1234 + rank, device=dev)
    else:
        batch_n, h, w = args.batch, args.height, args.width
        cfg = rcnet_main.ZJU_CONFIG
        model = rcnet_main.build_model(dev, cfg)
        main_mod, extra, stages = rcnet_main, {}, rcnet_stages(model)
        batch = rcnet_main.synthetic_batch(batch_n, h, w, cfg, seed=1234 + rank, device=dev)
    model.train()
    opt = FlatAdam(model.parameters(), lr=cfg['learning_rate'])
    reducer = GradientAllReducer(opt, stages=stages, mode=args.allreduce, comm=args.rccl_comm) if (world > 1 or args.force_ddp) else None
    if reducer is not None:
        reducer.broadcast_parameters(0)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    extra["loss_scale"] = args.loss_scale if args.loss_scale is not None else (16384.0 if args.dtype == "fp16" else 1.0)
    if args.eager:
        def step():
            return main_mod.train_step(model, opt, batch, cfg, reducer, **extra)
    else:
        try:
            step = main_mod.GraphedTrainStep(model, opt, batch, cfg, reducer, **extra)
        except Exception as ex:
            if reducer is None or reducer.comm is None:
                raise
            # the captured-collective step could not be built (every rank runs the same code on the same shapes, so every rank lands here):
            # fall back to torch.distributed's collectives between per-stage graphs rather than lose the measurement
            sys.stderr.write("bench.py: rank %d: single-graph step with captured RCCL buckets failed (%r); falling back to --comm torch\n" % (rank, ex))
            torch.cuda.synchronize()
            reducer.close()
            args.rccl_comm = None
            reducer = GradientAllReducer(opt, stages=stages, mode=args.allreduce)
            step = main_mod.GraphedTrainStep(model, opt, batch, cfg, reducer, **extra)
    for _ in range(warmup):
        loss = step()
    settle = 0
    if args.settle_seconds > 0:      # untimed: bring clocks / power state to the steady state the timed steps then run in
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go = True
        while go:
            for _ in range(5):
                loss = step()
            torch.cuda.synchronize()
            settle += 5
            go = time.perf_counter() - t0 < args.settle_seconds
            if world > 1:    # rank 0 decides, so that every rank runs the same number of steps (and collectives)
                import torch.distributed as dist
                flag = torch.tensor([1 if go else 0], dtype=torch.int32, device=dev)
                dist.broadcast(flag, 0)
                go = bool(flag.item())
    timer = engine.KernelTimer(repeat=1 if args.eager else args.timer_repeat)
    if args.eager:
        engine.set_kernel_timer(timer)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    engine.set_kernel_timer(None)
    final_loss = float(loss.detach()) if loss is not None else float('nan')
    engine.check_roi_overflow()      # raises if a timed step met a RoI geometry the one-byte arg-max cannot encode (its gradients would be NaN)
    timed_steps = steps
    if not args.eager:
        # per-kernel HIP-event timing cannot run inside graph replays: the same step is re-run eagerly (same kernels,
        # same shapes, same stream) for a few instrumented iterations right after the timed region
        timed_steps = min(3, max(1, steps))
        # one untimed eager step first: the eager path allocates tensors the graph's private pool never handed to the caching allocator, and a
        # hipMalloc inside an event pair was seen as a 0.9-ms "kernel" (configs[4] leg, 18 x the same launch's duration in four other runs)
        main_mod.train_step(model, opt, batch, cfg, reducer, **extra)
        torch.cuda.synchronize()
        engine.set_kernel_timer(timer)
        for _ in range(timed_steps):
            main_mod.train_step(model, opt, batch, cfg, reducer, **extra)
        torch.cuda.synchronize()
        engine.set_kernel_timer(None)
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if reducer is not None:
        reducer.close()
    ms = elapsed * 1e3 / max(steps, 1)
    out = dict(value=batch_n * world * steps / elapsed, ms_per_step=ms, steps=steps, warmup=warmup, settle_steps=settle, final_loss=final_loss,
               batch_per_gpu=batch_n, height=h, width=w,
               launch_mode="eager" if args.eager else (
                   "one hipGraph (fwd+bwd+captured RCCL buckets) + eager Adam" if (reducer is not None and reducer.comm is not None) else
                   "hipGraphs split at the stage marks (fwd+bwd) + eager all-reduce/Adam" if reducer is not None else "one hipGraph (fwd+bwd) + eager Adam"))
    if rank == 0:
        key = "%s_b%d_%dx%d_%s" % (kind, batch_n, h, w, args.dtype)
        out["roofline"] = kernel_roofline(timer, timed_steps, ms, args.dtype, key)
        out["roofline_conv"] = kernel_roofline(timer, timed_steps, ms, args.dtype, key, conv_only=True, with_tables=False)
        if args.detail:
            rows = sorted(timer.detail().items(), key=lambda kv: -kv[1][1])
            with open(args.detail + ("" if (kind == "rcnet" and not override) else "." + "_".join([kind] + [str(v) for k, v in sorted(override.items()) if k != "settle_seconds"])), "w") as f:
                for (k, desc), (n, tms, fl, by) in rows:
                    f.write("%-12s %-48s launches/step %5.1f  ms/step %8.3f  TFLOP/s %7.2f  GB/s(alg) %8.1f\n" % (
                        k, desc, n / timed_steps, tms / timed_steps, fl / (tms * 1e-3) / 1e12 if tms > 0 else 0.0, by / (tms * 1e-3) / 1e9 if tms > 0 else 0.0))
    del step, opt, model
    engine.set_param_grad_allocator(None)
    engine.clear_caches()
    torch.cuda.empty_cache()
    return out


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
    if have < n:
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
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE configs[1]: 8)")
    ap.add_argument("--config3", action="store_true", help="BASELINE configs[3]: global batch 32 over the ranks (4 images per rank on 8 GPUs)")
    ap.add_argument("--dtype", default=os.environ.get("RIDERS_BENCH_DTYPE", "bf16"), choices=["fp32", "bf16", "fp16"],
                    help="activation dtype (BASELINE.json configs[1] quotes bf16; fp32 is the 1e-3 parity mode; fp16 = configs[4], with --height 512 "
                         "--width 1024, static loss scale --loss-scale)")
    ap.add_argument("--loss-scale", type=float, default=None, help="static loss scale (default 16384 for fp16, 1 otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--workload", default="rcnet", choices=["rcnet", "sml"],
                    help="headline workload: rcnet = BASELINE configs[1]; sml = configs[2] alone")
    ap.add_argument("--no-sml", action="store_true", help="skip the second (SML, configs[2]) entry of the default N = 1 run")
    ap.add_argument("--no-legs", action="store_true", help="skip the fp32 (parity mode) and configs[4] (fp16 512x1024) entries of the default N = 1 run")
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
    ap.add_argument("--no-live-traffic", action="store_true", help="do not start the two rocprofv3 counter passes (FETCH_SIZE / WRITE_SIZE children, ~1 min) "
                    "that measure roofline.traffic inside the default N = 1 run; the committed profiles/r05_traffic.json values are reported instead")
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
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the riders_amd hot path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    ddp = world > 1 or args.force_ddp
    comm = None
    if ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        assert dist.get_world_size() == world
        comm = dict(backend="nccl (RCCL)", world_size=dist.get_world_size(), rccl_version=".".join(str(v) for v in torch.cuda.nccl.version()))
    args.rccl_comm = None
    if ddp and args.comm == "c_abi":
        from riders_amd.parallel import RcclComm
        import torch.distributed as dist
        ok = torch.ones(1, device=dev)
        try:
            args.rccl_comm = RcclComm(rank, world)
        except Exception as ex:      # e.g. no loadable librccl.so: fall back to torch.distributed's collectives on EVERY rank (decided together below)
            sys.stderr.write("bench.py: rank %d: rd_comm_init failed (%r)\n" % (rank, ex))
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            if args.rccl_comm is not None:
                args.rccl_comm.close()
            args.rccl_comm = None
            sys.stderr.write("bench.py: falling back to --comm torch\n")
        comm["transport"] = "rd_comm (C ABI, captured)" if args.rccl_comm is not None else "torch.distributed"

    if args.opts:
        from riders_amd import engine
        engine.apply_opts(args.opts)
    head = run_workload(args.workload, args, dev, world, rank, args.steps, args.warmup)
    sml = None
    legs = {}
    default_line = args.workload == "rcnet" and world == 1 and args.dtype == "bf16" and (args.height, args.width, args.batch) == (256, 512, 8) \
        and not args.config3 and not args.force_ddp and not args.eager
    if args.workload == "rcnet" and world == 1 and not args.no_sml:
        sml = run_workload("sml", args, dev, world, rank, args.steps, args.warmup, settle_seconds=min(args.settle_seconds, 1.0))
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
    if rank == 0:
        cpu = val = None
        if default_line and not args.no_live_traffic:
            roofs = [head.get("roofline"), head.get("roofline_conv")]
            live = live_traffic(args, set(r["kernel"] for r in roofs if r and r.get("kernel")))
            for r in roofs:
                apply_live_traffic(r, live)
        if world == 1 and not args.no_cpu_baseline and args.workload == "rcnet":
            try:
                val = val_abs_rel_pair(dev)
            except Exception as ex:      # never lose the bench line to the auxiliary figure
                val = dict(error=repr(ex)[:300])
            cpu = cpu_baseline()
        full = full_record(args, world, comm, ddp, head, sml, legs, cpu, val)
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
