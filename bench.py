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

Prints ONE JSON line on rank 0 (see the contract in the task statement) including
  "roofline":     the kernel FAMILY with the largest share of the step (HIP events on the launch stream around every launch of an
                  instrumented eager run of the same step): algorithmic FLOPs / bytes over summed launch time against the roof
                  that binds it, the time-weighted fraction of every family, and PMC traffic from profiles/r02_traffic.json
  "cpu_baseline": the oracle (PyTorch-CPU restatement of the reference path) timed on this box's host cores on a
                  bounded sample (test infrastructure used as the reported baseline only)
  "sml":          the same measurement for BASELINE.json configs[2] (Scale Map Learner, batch 16, bf16) at 256x512 (N = 1 only).
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
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r02_traffic.json")   # written by tools/traffic_pass.sh (rocprofv3 --pmc passes)


def cpu_baseline(seconds_budget=25.0):
    """Oracle full training step (fwd + loss + bwd + Adam) with torch CPU ops on all host threads; B=1 (K=30, 256x512); imgs/s."""
    from oracle import rcnet as O
    from riders_amd import rcnet_main
    cfg = rcnet_main.ZJU_CONFIG
    torch.manual_seed(0)
    model = rcnet_main.build_model(torch.device('cpu'), cfg)
    sd_e = {k: v.detach().clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in model.encoder.state_dict().items()}
    sd_d = {k: v.detach().clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in model.decoder.state_dict().items()}
    leaves = [v for d in (sd_e, sd_d) for v in d.values() if v.requires_grad]
    state = [(torch.zeros_like(v), torch.zeros_like(v)) for v in leaves]
    img, pts, boxes, gt = rcnet_main.synthetic_batch(1, 256, 512, cfg, seed=99)
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
    t0 = time.time()
    step(); step()  # 2 warm-up steps
    first = (time.time() - t0) / 2
    n, t0, times = 0, time.time(), []
    while n < 3 or (time.time() - t0 < seconds_budget - 2 * first and n < 12):
        t1 = time.time()
        step()
        times.append(time.time() - t1)
        n += 1
    times.sort()
    return dict(value=1.0 / times[len(times) // 2], unit="imgs/s", cores=torch.get_num_threads(), kind="port", best=1.0 / times[0],
                sample="oracle RC-Net full step (fwd+loss+bwd+Adam), B=1 (30 ROIs, 256x512), fp32, median of %d timed steps after 2 warm-ups" % n)


def family_roofline(timer, timed_steps, ms_per_step, dtype, traffic_key):
    """Per-family table from the HIP-event records + the roofline object of the dominant family."""
    peak_mfma = PEAK_MFMA[dtype]
    fams = {}
    for (kind, desc), (n, tms, fl, by) in timer.detail().items():
        f = fams.setdefault(kind, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, t_roof_ms=0.0, t_mfma_ms=0.0, t_hbm_ms=0.0, shapes=[]))
        t_m, t_h = fl / (peak_mfma * 1e12) * 1e3, by / (PEAK_HBM * 1e9) * 1e3
        f["ms"] += tms; f["flops"] += fl; f["bytes"] += by; f["launches"] += n
        f["t_roof_ms"] += max(t_m, t_h); f["t_mfma_ms"] += t_m if t_m >= t_h else 0.0; f["t_hbm_ms"] += t_h if t_h > t_m else 0.0
        f["shapes"].append((tms, desc, n, fl, by))
    if not fams:
        return None
    table = {}
    for k, f in fams.items():
        bound = "mfma" if f["t_mfma_ms"] >= f["t_hbm_ms"] else "hbm"
        ach = f["flops"] / (f["ms"] * 1e-3) / 1e12 if bound == "mfma" else f["bytes"] / (f["ms"] * 1e-3) / 1e9
        peak = peak_mfma if bound == "mfma" else PEAK_HBM
        table[k] = dict(ms_per_step=f["ms"] / timed_steps, share_of_step=(f["ms"] / timed_steps) / ms_per_step, launches_per_step=f["launches"] / timed_steps,
                        bound=bound, achieved=ach, peak=peak, unit="TFLOP/s" if bound == "mfma" else "GB/s", frac=ach / peak,
                        frac_time_weighted=f["t_roof_ms"] / f["ms"] if f["ms"] > 0 else 0.0)
    dom = max(fams, key=lambda k: fams[k]["ms"])
    f, row = fams[dom], table[dom]
    top = sorted(f["shapes"], reverse=True)[:3]
    roof = dict(bound=row["bound"], kernel="family %s (%d launches/step)" % (dom, round(row["launches_per_step"])), achieved=row["achieved"],
                peak=row["peak"], unit=row["unit"], frac=row["frac"], traffic=None, frac_time_weighted=row["frac_time_weighted"],
                share_of_step=row["share_of_step"], ms_per_step=row["ms_per_step"], avg_launch_us=f["ms"] * 1e3 / f["launches"],
                algorithmic_flops_per_launch=f["flops"] / f["launches"], algorithmic_bytes_per_launch=f["bytes"] / f["launches"],
                top_shapes=[dict(shape=d, ms_per_step=t / timed_steps, tflops=fl / (t * 1e-3) / 1e12 if t > 0 else 0.0,
                                 gbs=by / (t * 1e-3) / 1e9 if t > 0 else 0.0) for t, d, n, fl, by in top],
                families=table,
                note="dominant = family with the largest summed launch time; achieved = algorithmic FLOPs (2/MAC) or bytes (each operand once) of "
                     "ALL its launches / their summed HIP-event durations (events on the launch stream, %d instrumented eager steps of the same "
                     "workload right after the timed region); frac_time_weighted prices every launch shape against the roof that binds IT "
                     "(HBM 8 TB/s or dense MFMA)" % timed_steps)
    try:    # HBM bytes per launch from PMC counters: separate rocprofv3 passes of the same command (tools/traffic_pass.sh), per family
        tr = json.load(open(TRAFFIC_FILE)).get(traffic_key, {}).get(dom)
        if tr is not None:
            roof["traffic"] = tr["bytes_per_step"] / max(row["launches_per_step"], 1e-9)
            roof["traffic_over_algorithmic"] = tr["bytes_per_step"] / max((f["bytes"] / timed_steps), 1.0)
            roof["traffic_source"] = tr.get("note", "")
            # the counters come from separate rocprofv3 passes (they cannot be collected inside this run): say so when the kernel sources
            # have changed since those passes
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from traffic_aggregate import csrc_fingerprint
            roof["traffic_stale"] = json.load(open(TRAFFIC_FILE)).get(traffic_key, {}).get("_csrc_sha1") != csrc_fingerprint(ROOT)
    except (OSError, ValueError):
        pass
    return roof


def run_workload(kind, args, dev, world, rank, steps, warmup):
    from riders_amd import engine, rcnet_main, sml_main
    from riders_amd.optim import FlatAdam
    from riders_amd.parallel import GradientAllReducer, rcnet_stages, sml_stages
    engine.set_compute_dtype(args.dtype)
    engine.clear_caches()
    torch.manual_seed(0)  # identical initial weights on every rank
    if kind == "sml":
        batch_n, h, w = args.sml_batch, args.sml_height, args.sml_width
        cfg = sml_main.ZJU_SML_CONFIG
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
    reducer = GradientAllReducer(opt, stages=stages) if (world > 1 or args.force_ddp) else None
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
    timer = engine.KernelTimer()
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
    timed_steps = steps
    if not args.eager:
        # per-kernel HIP-event timing cannot run inside graph replays: the same step is re-run eagerly (same kernels,
        # same shapes, same stream) for a few instrumented iterations right after the timed region
        timed_steps = min(3, max(1, steps))
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
               launch_mode="eager" if args.eager else ("hipGraphs split at the stage marks (fwd+bwd) + eager all-reduce/Adam" if reducer is not None
                                                       else "one hipGraph (fwd+bwd) + eager Adam"))
    if rank == 0:
        out["roofline"] = family_roofline(timer, timed_steps, ms, args.dtype, "%s_b%d_%dx%d_%s" % (kind, batch_n, h, w, args.dtype))
        if args.detail:
            rows = sorted(timer.detail().items(), key=lambda kv: -kv[1][1])
            with open(args.detail if kind == "rcnet" else args.detail + ".sml", "w") as f:
                for (k, desc), (n, tms, fl, by) in rows:
                    f.write("%-12s %-48s launches/step %5.1f  ms/step %8.3f  TFLOP/s %7.2f  GB/s(alg) %8.1f\n" % (
                        k, desc, n / timed_steps, tms / timed_steps, fl / (tms * 1e-3) / 1e12 if tms > 0 else 0.0, by / (tms * 1e-3) / 1e9 if tms > 0 else 0.0))
    del step, opt, model
    engine.set_param_grad_allocator(None)
    engine.clear_caches()
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-seconds", type=float, default=2.0, help="untimed replays before the timed region (clock settling)")
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE configs[1]: 8)")
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
    ap.add_argument("--sml-batch", type=int, default=16)
    ap.add_argument("--sml-height", type=int, default=256)
    ap.add_argument("--sml-width", type=int, default=512)
    ap.add_argument("--eager", action="store_true", help="do not capture forward+backward into hipGraphs")
    ap.add_argument("--force-ddp", action="store_true", help="run the N > 1 code path (RCCL process group, stage-bucketed all-reduce) on however "
                                                              "many ranks there are, including one: a functional check of that path on a 1-GPU box")
    ap.add_argument("--detail", default=None, help="write a per-launch-shape timing table to this file")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the riders_amd hot path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    ddp = world > 1 or args.force_ddp
    if ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)

    head = run_workload(args.workload, args, dev, world, rank, args.steps, args.warmup)
    sml = None
    if args.workload == "rcnet" and world == 1 and not args.no_sml:
        sml = run_workload("sml", args, dev, world, rank, args.steps, args.warmup)
    if rank == 0:
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
                "note": "BASELINE configs[1] per rank at every N (weak scaling); configs[3]'s global 32 on 8 GPUs is --batch 4"},
            "final_loss": head["final_loss"], "launch_mode": head["launch_mode"], "settle_steps": head["settle_steps"],
            "allreduce": None if world == 1 else "RCCL sum of the flat fp32 gradient arena in 3 stage buckets, each started when the backward "
                                                 "passes its stage mark (overlaps the remaining backward graphs); 1/N folded into Adam",
            "roofline": head["roofline"],
        }
        if sml is not None:
            out["sml"] = {"metric": "train imgs/sec (Scale Map Learner, MiDaS-small / EfficientNet-Lite3; BASELINE configs[2])",
                          "value": sml["value"], "unit": "imgs/s", "ms_per_step": sml["ms_per_step"], "dtype": out["dtype"],
                          "config": {"workload": "SML training step, batch %d/GPU, %dx%d frames, device pre-step+fwd+loss+bwd+Adam" % (
                              sml["batch_per_gpu"], sml["height"], sml["width"])},
                          "final_loss": sml["final_loss"], "settle_steps": sml["settle_steps"], "roofline": sml["roofline"]}
        if world == 1 and not args.no_cpu_baseline and is_rc:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if ddp:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
