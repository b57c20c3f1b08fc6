"""Host-side logic on the GPU-less build container (kernels run through the tests/emu host build of the same sources):
optimizer state interchange with torch.optim.Adam, skipped (gradient-less) parameters, in-place re-packing of cached MFMA
operands, and the staged backward (engine.StepTape + stage marks) that the split hipGraph capture and the overlapped gradient
all-reduce are built on."""
import numpy as np
import torch

from tests.parity_cases import close


def _toy_params(seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(7, 5, generator=g)), torch.nn.Parameter(torch.randn(33, generator=g)),
            torch.nn.Parameter(torch.randn(4, 3, 3, 3, generator=g))]


def _grads(step, shapes, seed=11):
    g = torch.Generator().manual_seed(seed + step)
    return [torch.randn(s, generator=g) for s in shapes]


def test_flat_adam_matches_torch_and_exchanges_state(emu):
    """RCNet/rcnet_model.py:224-257: `radarnet_optimizer_state_dict` is torch.optim.Adam's state_dict.  FlatAdam must produce that
    layout, accept it, and continue bit-compatibly (1e-6) in both directions."""
    from riders_amd.optim import FlatAdam
    ours, theirs = _toy_params(), _toy_params()
    shapes = [p.shape for p in ours]
    fa, ta = FlatAdam(ours, lr=2e-4), torch.optim.Adam(theirs, lr=2e-4)
    for s in range(3):
        gs = _grads(s, shapes)
        fa.zero_grad()
        for p, q, g in zip(ours, theirs, gs):
            p.grad, q.grad = g.clone(), g.clone()
        fa.step(); ta.step()
    for p, q in zip(ours, theirs):
        close(p, q, 1e-6, "param after 3 steps")
    sd = fa.state_dict()
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][0]["params"] == [0, 1, 2]
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 3.0
    # ours -> torch
    fresh = _toy_params(5)
    with torch.no_grad():
        for q, p in zip(fresh, ours):
            q.copy_(p)
    tb = torch.optim.Adam(fresh, lr=1e-3)
    tb.load_state_dict(sd)
    assert tb.param_groups[0]["lr"] == 2e-4
    # torch -> ours
    fresh2 = _toy_params(6)
    with torch.no_grad():
        for q, p in zip(fresh2, theirs):
            q.copy_(p)
    fb = FlatAdam(fresh2, lr=1e-3)
    fb.load_state_dict(ta.state_dict())
    assert fb.param_groups[0]["lr"] == 2e-4 and fb.steps == [3, 3, 3]
    gs = _grads(9, shapes)
    fb.zero_grad()
    for p, q, r, g in zip(fresh2, fresh, theirs, gs):
        p.grad, q.grad, r.grad = g.clone(), g.clone(), g.clone()
    fb.step(); tb.step(); ta.step()
    for p, q, r in zip(fresh2, fresh, theirs):
        close(p, r, 1e-6, "torch state loaded into FlatAdam")
        close(q, r, 1e-6, "FlatAdam state loaded into torch")
    # round trip through torch.save
    import io
    buf = io.BytesIO()
    torch.save(fb.state_dict(), buf); buf.seek(0)
    fc = FlatAdam(_toy_params(7), lr=1.0)
    fc.load_state_dict(torch.load(buf))
    assert torch.equal(fc.exp_avg, fb.exp_avg) and torch.equal(fc.exp_avg_sq, fb.exp_avg_sq) and fc.steps == fb.steps


def test_flat_adam_skips_parameters_without_gradient(emu):
    """torch skips a parameter whose grad is None (no moment decay, no step increment); a parameter that is used only every other
    step must follow torch exactly, and a never-used one (the reference's idle `projection` convolutions) has no optimizer state."""
    from riders_amd.optim import FlatAdam
    ours, theirs = _toy_params(), _toy_params()
    shapes = [p.shape for p in ours]
    fa, ta = FlatAdam(ours, lr=1e-2), torch.optim.Adam(theirs, lr=1e-2)
    for s in range(4):
        gs = _grads(s, shapes)
        fa.zero_grad(); ta.zero_grad()
        for i, (p, q, g) in enumerate(zip(ours, theirs, gs)):
            if i == 2 or (i == 1 and s % 2 == 1):     # param 2 never trains, param 1 only on even steps
                continue
            view = fa._grad_view(p)                    # what a backward kernel does: write the arena slot
            view.copy_(g)
            p.grad, q.grad = view, g.clone()
        fa.step(); ta.step()
    for p, q in zip(ours, theirs):
        close(p, q, 1e-6, "intermittently used parameter")
    assert fa.steps == [4, 2, 0]
    sd = fa.state_dict()
    assert sorted(sd["state"]) == [0, 1] and sorted(ta.state_dict()["state"]) == [0, 1]


def test_flat_adam_replay_marking_and_guarded_fp16_step(emu):
    """(1) A hipGraph replay writes the gradient arena without passing the gradient allocator; with `zero_grad()` between replays (the
    reference loop's habit) the step must still update: GraphedStep re-marks the slots its capture pass touched (`mark_touched`).
    (2) fp16 mode (static loss scale): a non-finite scaled gradient skips the whole update on the device and is counted."""
    from riders_amd.optim import FlatAdam
    ours, theirs = _toy_params(), _toy_params()
    shapes = [p.shape for p in ours]
    fa, ta = FlatAdam(ours, lr=1e-2), torch.optim.Adam(theirs, lr=1e-2)
    gs = _grads(0, shapes)
    for p, g in zip(ours, gs):
        fa._grad_view(p).copy_(g)                      # "capture pass": the backward kernels write the arena through the allocator
    touched = fa.touched_indices()
    assert touched == [0, 1, 2]
    fa.zero_grad()                                     # the training loop clears the flags ...
    before = [p.detach().clone() for p in ours]
    fa.step()                                          # ... and without re-marking nothing is live: torch semantics, no update
    assert all(torch.equal(p, b) for p, b in zip(ours, before))
    fa.mark_touched(touched)                           # what GraphedStep.__call__ does after its replays
    fa.step()
    for q, g in zip(theirs, gs):
        q.grad = g.clone()
    ta.step()
    for p, q in zip(ours, theirs):
        close(p, q, 1e-6, "replayed step after zero_grad")
    # (2) guarded step
    fa.loss_scale = 1024.0
    fa.zero_grad()
    for p, g in zip(ours, _grads(1, shapes)):
        fa._grad_view(p).copy_(g * 1024.0)
    fa.flat_grad[5] = float("inf")
    before = [p.detach().clone() for p in ours]
    m_before = fa.exp_avg.clone()
    fa.step()
    assert all(torch.equal(p, b) for p, b in zip(ours, before)) and torch.equal(fa.exp_avg, m_before), "overflow must skip the update"
    assert fa.skipped_steps() == 1
    fa.zero_grad()
    for p, g in zip(ours, _grads(1, shapes)):
        fa._grad_view(p).copy_(g * 1024.0)
    fa.step()
    assert fa.skipped_steps() == 1 and not all(torch.equal(p, b) for p, b in zip(ours, before))


def test_cast_rejects_mixed_half_types(emu):
    """bf16 <-> fp16 would run as a bit copy inside one precision build (rd_api.cpp): the entry points refuse the pair."""
    from riders_amd import engine
    from riders_amd._lib import RD_BF16, RD_F16, RD_F32
    lib = engine.L()
    a, b = torch.zeros(8, dtype=torch.bfloat16), torch.zeros(8, dtype=torch.float16)
    assert lib.rd_cast(engine._p(a), engine._p(b), 8, RD_BF16, RD_F16, 1.0, None) != 0
    assert lib.rd_cast(engine._p(b), engine._p(a), 8, RD_F16, RD_BF16, 1.0, None) != 0
    assert b"fp16" in lib.rd_last_error_string()
    assert lib.rd_nchw_to_nhwc(engine._p(a), engine._p(b), 1, 2, 2, 2, RD_BF16, RD_F16, 1.0, None) != 0
    f = torch.arange(8, dtype=torch.float32)
    assert lib.rd_cast(engine._p(f), engine._p(b), 8, RD_F32, RD_F16, 1.0, None) == 0 and torch.equal(b.float(), f)


def test_bench_launcher_argument_checks():
    """bench.py --gpus N: WORLD_SIZE must equal --gpus (exit 2); without WORLD_SIZE it spawns its ranks itself and refuses when fewer
    GPUs are visible (here: none) -- before anything touches a GPU."""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and "only 0 GPU" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0"), capture_output=True,
                       text=True)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr


def test_packed_operand_is_repacked_in_place(emu):
    """A weight written through torch (load_state_dict / broadcast) keeps its packed MFMA operand at the same address (captured
    hipGraphs read it) and the operand follows the new values."""
    from riders_amd import engine, net_utils
    m = net_utils.Conv2d(8, 8, 3, 1, 'kaiming_uniform', None, False)
    x = torch.randn(1, 8, 6, 5)
    with torch.no_grad():
        y0 = m(x)
    ent = engine._pack_cache[id(m.conv.weight)]
    (slot, (key0, buf0)), = [(k, v) for k, v in ent.items() if k != "ref"]
    addr = buf0.data_ptr()
    new_w = torch.randn_like(m.conv.weight)
    m.load_state_dict({"conv.weight": new_w})
    engine.refresh_packed()
    assert ent[slot][1].data_ptr() == addr and ent[slot][0] != key0
    with torch.no_grad():
        y1 = m(x)
    assert ent[slot][1].data_ptr() == addr
    ref = torch.nn.functional.conv2d(x, new_w, padding=1)
    close(y1, ref, 1e-3, "forward after in-place re-pack")
    assert float((y1 - y0).abs().max()) > 1e-3
    # without the explicit refresh the next forward re-packs into the same buffer as well
    with torch.no_grad():
        m.conv.weight.copy_(torch.randn_like(new_w))
        y2 = m(x)
    assert ent[slot][1].data_ptr() == addr
    close(y2, torch.nn.functional.conv2d(x, m.conv.weight.detach(), padding=1), 1e-3, "forward after torch write")


class _TwoStage(torch.nn.Module):
    """Decoder-style toy with one stage mark: body (Conv2d+BN+LReLU, 1x1 projection) | head (DecoderBlock + 1-channel output conv)."""

    def __init__(self):
        super().__init__()
        from riders_amd import net_utils
        act = net_utils.activation_func('leaky_relu')
        self.body = net_utils.Conv2d(4, 8, 3, 1, 'kaiming_uniform', act, True)
        self.proj = net_utils.Conv2d(8, 8, 1, 1, 'kaiming_uniform', None, False)
        self.head = net_utils.DecoderBlock(8, 0, 8, 'kaiming_uniform', act, True)
        self.out = net_utils.Conv2d(8, 1, 3, 1, 'kaiming_uniform', None, False)

    def forward(self, x):
        from riders_amd import engine

        def run(x):
            h = self.proj._fwd(self.body._fwd(engine.from_nchw(x)))
            engine.stage_mark("head_done")
            h = self.out._fwd(self.head._fwd(h, shape=(2 * x.shape[2], 2 * x.shape[3])))
            return engine.to_nchw_out(h, torch.float32)
        return engine.run_region(run, (x,), list(self.parameters()))


def _toy_loss(model, x, label, valid):
    from riders_amd import engine
    logits = model(x)
    return engine.run_region(lambda lg: engine.bce_masked(lg, label, valid, 2.5), (logits,), [])


def _toy_batch(seed, n=2):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 4, 6, 5, generator=g)
    label = (torch.rand(n, 1, 12, 10, generator=g) > 0.5).float()
    valid = (torch.rand(n, 1, 12, 10, generator=g) > 0.3).float()
    return x, label, valid


def test_staged_backward_equals_autograd(emu):
    """engine.StepTape (one tape, driven stage by stage from the calling thread) produces the gradients torch.autograd produces
    through the module's region, stops at every stage mark, and issues the grouped 1x1 weight gradients before the mark's hooks."""
    from riders_amd import engine, rcnet_main
    from riders_amd.optim import FlatAdam
    from riders_amd.parallel import GradientAllReducer
    torch.manual_seed(3)
    model = _TwoStage(); model.train()
    opt = FlatAdam(list(model.parameters()), lr=1e-3)
    x, label, valid = _toy_batch(1)
    loss = _toy_loss(model, x, label, valid)
    opt.zero_grad(); loss.backward()
    ref = opt.flat_grad.clone()
    assert float(ref.abs().max()) > 0
    opt.flat_grad.zero_()
    red = GradientAllReducer(opt, stages={"head_done": list(model.head.parameters()) + list(model.out.parameters())})
    try:
        tags = []
        loss2 = rcnet_main.staged_gradients(lambda: _toy_loss(model, x, label, valid), opt, tags.append)
        assert tags == ["head_done"]
        assert float(loss2) == float(loss)
        assert torch.equal(opt.flat_grad, ref)
        # the stage hook started the head bucket at the mark; reduce() adds the rest of the arena, front to back
        assert [t for t, _, _ in red.log] == ["head_done"]
        red.reduce()
        issued = sorted((s, e) for _, s, e in red.log)
        assert issued[0][0] == 0 and issued[-1][1] == opt.numel and all(a[1] == b[0] for a, b in zip(issued, issued[1:]))
        head_lo = min(opt.offsets[opt._index[id(p)]] for p in list(model.head.parameters()) + list(model.out.parameters()))
        assert red.log[0][1] == head_lo
    finally:
        red.close()
    assert all(p.grad is not None for p in model.parameters())


def test_stage_ranges_cover_rcnet_arena():
    """parallel.rcnet_stages: decoder | transformer + point MLP | (rest = image encoder) are contiguous arena slices."""
    from riders_amd import rcnet_main
    from riders_amd.optim import FlatAdam
    from riders_amd.parallel import GradientAllReducer, rcnet_stages
    cfg = dict(rcnet_main.ZJU_CONFIG, patch_size=[32, 32], n_filters_encoder_image=[8, 8, 16, 16, 128],
               n_neurons_encoder_depth=[8, 8, 8, 8, 128], n_filters_decoder=[16, 16, 8, 8, 8])
    model = rcnet_main.build_model(torch.device('cpu'), cfg)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    red = GradientAllReducer(opt, stages=rcnet_stages(model))
    try:
        assert len(red.stage_ranges["decoder_done"]) == 1 and len(red.stage_ranges["attention_done"]) == 1
        n_img = sum((p.numel() + 3) // 4 * 4 for p in model.encoder.encoder_image.parameters())
        assert red.rest == [(0, n_img)]
        assert red.stage_ranges["attention_done"][0][0] == n_img
        assert red.stage_ranges["decoder_done"][0] == (red.stage_ranges["attention_done"][0][1], opt.numel)
    finally:
        red.close()
        from riders_amd import engine
        engine.set_param_grad_allocator(None)


def test_bench_line_stays_under_the_driver_limit():
    """the 39.8-KB stdout line was not parsed.  The compact line built from a canned kernel timer with many kernels, families and
    launch shapes, every leg present and long notes must stay under 6000 bytes, carry the contract's fields and parse back."""
    import argparse
    import importlib.util
    import json
    import os
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from riders_amd.engine import KernelTimer

    class Ev(object):
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    timer = KernelTimer(repeat=5)
    names = ["conv3x3_frag_kernel<rd::bf16_t, 4, 2, 4, true, true, false>", "conv3x3_wgrad_tr_kernel<4, 2, 16, 4, false>",
             "bn_bwd_apply_vec_kernel<rd::bf16_t, true, 2>", "col_reduce_vec_kernel<rd::bf16_t, true, 2>"] + \
            ["conv_gemm_kernel<rd::bf16_t, %d, true, 2, true, false>" % i for i in range(40)]
    for i, n in enumerate(names):
        kind = "conv_gemm" if "conv" in n and "wgrad" not in n else ("conv_wgrad" if "wgrad" in n else "bn_backward")
        for j in range(12):
            timer.records.setdefault(kind, []).append((Ev(0.0), Ev(0.05 + 0.001 * i + 0.0001 * j), 4e10 if "conv" in n else 0.0,
                                                       "layer %d shape M=%d Cin=128 Cout=256 with a long description string %s" % (i, 1000 * j, "x" * 60),
                                                       1e8, n, 5))
    prof = {n: (4.0, 4 * 52000.0) for n in names}      # what the kernel-trace child returns: (calls per step, ns per step)
    live = {n: dict(bytes_per_launch=9.0e7, dispatches=12, mfma_util=0.31, mfma_busy_cycles=4.0e7, gui_active=1.0e6) for n in names}
    roofs = bench.build_rooflines(timer, 3, 7.4, "bf16", prof, live, "rcnet_b8_256x512_bf16")
    assert "kernels" in roofs and "families" in roofs and len(json.dumps(roofs)) > 4000
    dom = roofs["roofline"]
    assert dom["kernel"] == "conv_gemm_kernel" and dom["instantiations"] == 40 and dom["what"] == "kernel family"      # the FAMILY with the largest summed time
    assert abs(dom["avg_launch_us"] - 52.0) < 1e-6 and "rocprofv3" in dom["duration_source"] and dom["traffic_live"] and abs(dom["mfma_util"] - 40.0 / 128.0) < 1e-9
    assert abs(dom["frac"] - dom["achieved"] / dom["peak"]) < 1e-12
    assert roofs["roofline_kernel"]["what"] == "kernel instantiation"
    no_prof = bench.build_rooflines(timer, 3, 7.4, "bf16", None, None, "k")["roofline"]
    assert "HIP events" in no_prof["duration_source"] and no_prof["traffic"] is None
    leg = dict(value=1089.123456789, ms_per_step=7.3456789, steps=200, warmup=10, settle_steps=250, final_loss=0.69314718, batch_per_gpu=8, height=256,
               width=512, launch_mode="one hipGraph (fwd+bwd) + eager Adam", launches_per_step=352.0, kernel_ms_per_step=6.9, **roofs)
    chain = dict(value=601.5, ms_per_step=26.6, steps=200, warmup=10, settle_steps=80, final_loss=0.5, images_per_step=16, batch_per_gpu=16, height=256, width=512,
                 launch_mode="per stage: one hipGraph (fwd+bwd) + eager Adam")
    args = argparse.Namespace(workload="chain", steps=200, warmup=10, dtype="bf16", config3=False, allreduce="all_reduce")
    cpu = dict(value=1.61745833, unit="imgs/s", cores=16, kind="port", host_cpus=256, thread_sweep_b1={"8": 1.4, "16": 1.6, "32": 1.2},
               sample="oracle RC-Net full step (fwd+loss+bwd+Adam), B=1 (30 ROIs, 256x512), fp32, best of 2 timed steps after 1 warm-up per thread count",
               rcnet_b8=dict(value=1.2, unit="imgs/s", cores=16, sample="one B=8 step"), chained=dict(value=0.9, unit="imgs/s"), seconds=13.4)
    val = dict(oracle=0.379123, sample="8 synthetic 256x512 frames (network input 288x576), random-init weights identical on every path",
               fp32=dict(hip=0.379146, diff_of_means=2.3e-5, max_abs_diff=2.7e-5), bf16=dict(hip=0.3801, diff_of_means=1e-3, max_abs_diff=2e-3), hip=0.379146, max_abs_diff=2.7e-5)
    caller = dict(rcnet=dict(value=400.0, ms_per_step=20.0, launches_per_step=1900.0, graphed_value=1100.0, launch_mode="eager", eager_value=350.0, eager_ms_per_step=23.0,
                             eager_launches_per_step=2100.0),
                  sml=dict(value=500.0, ms_per_step=32.0, launches_per_step=3000.0, graphed_value=1290.0))
    full = bench.full_record(args, 1, dict(backend="nccl (RCCL)", world_size=1, rccl_version="2.26.6"), True, chain, leg, dict(leg, batch_per_gpu=16),
                             {"fp32": leg, "config4": leg, "config4_sml": leg}, cpu, val, caller)
    assert len(json.dumps(full)) > 15000          # the kind of record round 4 printed on stdout
    text = bench.render_line(bench.compact_line(full))
    assert len(text) < 6000 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "rcnet", "sml", "roofline_families", "unchanged_caller", "val_abs_rel"):
        assert k in line, k
    assert line["metric"] == "train imgs/sec (RC-Net+SML, 256x512)"       # BASELINE.json's metric, word for word
    assert line["unit"] == "imgs/s" and "RC-Net" in line["config"]["workload"] and "SML" in line["config"]["workload"] and "model" not in line["config"]
    assert line["config"]["global_batch"] == 16
    r = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_launch_us", "avg_launch_us_hip_events"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "kernels" not in r and "families" not in r and "shapes" not in r
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    for name in ("rcnet", "sml", "fp32", "config4", "config4_sml"):
        assert set(line[name]) <= {"value", "unit", "ms_per_step", "dtype", "config", "roofline", "launches_per_step"} and set(line[name]["roofline"]) <= {"kernel", "bound", "frac"}
    assert set(line["val_abs_rel"]) == {"oracle", "frames", "fp32", "bf16"} and line["unchanged_caller"]["rcnet"]["launches_per_step"] == 1900.0
    # a pathological field cannot lose the line: optional blocks are dropped instead
    big = dict(bench.compact_line(full), comm={"x": "y" * 9000})
    assert len(bench.render_line(big)) < 6000 and "value" in json.loads(bench.render_line(big))
    # a run under a profiler starts no profiler of its own, and its children would not inherit the outer one's environment
    env = {"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so:/x/libfoo.so", "ROCP_TOOL_LIBRARIES": "x", "ROCPROF_OUTPUT_PATH": "/tmp/o", "PATH": "/bin"}
    assert bench.under_profiler(env) and not bench.under_profiler({"PATH": "/bin", "LD_PRELOAD": "/x/libfoo.so"})
    assert bench.clean_profiler_env(env) == {"LD_PRELOAD": "/x/libfoo.so", "PATH": "/bin"}


def test_c_abi_comm_loopback_and_bucketing(emu):
    """rd_comm_* through the emulator build's one-rank loop-back: RcclComm binds the entry points, GradientAllReducer(comm=...) issues the
    stage buckets through rd_allreduce_bucket (no torch.distributed), reduce() joins, and the error convention holds (negative = argument
    error with a message)."""
    import ctypes
    from riders_amd import engine
    from riders_amd.optim import FlatAdam
    from riders_amd.parallel import GradientAllReducer, RcclComm
    lib = engine.L()
    comm = RcclComm(rank=0, world=1)
    assert comm.world == 1 and comm.pending() == 0
    params = _toy_params()
    opt = FlatAdam(params, lr=1e-3)
    red = GradientAllReducer(opt, bucket_bytes=64, stages={"late": params[2:]}, comm=comm)
    try:
        assert red.collective and red.world == 1 and opt.grad_scale == 1.0
        red.broadcast_parameters(0)
        for p in params:
            opt._grad_view(p).copy_(torch.full_like(p, 2.0))
        red.on_stage("late")
        n_stage = len(red.log)
        assert n_stage >= 1 and all(t == "late" for t, _, _ in red.log) and comm.pending() == n_stage
        red.reduce()
        assert comm.pending() == 0 and len(red.log) == len(red.buckets)
        assert all(bool(torch.all(opt._grad_view(p) == 2.0)) for p in params)      # a sum over one rank
    finally:
        red.close()
    h = ctypes.c_void_p()
    idb = ctypes.create_string_buffer(128)
    assert lib.rd_comm_init(2, 2, idb, ctypes.byref(h)) < 0 and b"world" in lib.rd_last_error_string()      # rank outside the world
    assert lib.rd_comm_init(1, 2, idb, ctypes.byref(h)) < 0      # an id that rd_comm_unique_id never drew
    assert lib.rd_comm_available() == 0
    assert lib.rd_allreduce_bucket(comm.handle, None, 4, 0, None) < 0
    assert lib.rd_allreduce_bucket(comm.handle, engine._p(opt.flat_grad), 4, 7, None) < 0 and b"allreduce_bucket" in lib.rd_last_error_string()
    comm.close()


def test_committed_bench_line_is_reproducible_from_the_committed_kernel_trace():
    """the line's roofline.frac must follow from profiles/.  The committed line (profiles/r06_bench_default.json) and the committed
    rocprofv3 --kernel-trace --stats summary of the same run's child (profiles/r06_rcnet_b8_256x512_bf16_kernel_stats.csv): for each listed family
    frac == algorithmic work per launch / the CSV's average duration of that family / the roof, within 2 %."""
    import csv
    import json
    import os
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, os.path.join(root, "tools"))
    from traffic_aggregate import kernel_key
    line = json.loads(open(os.path.join(root, "profiles", "r06_bench_default.json")).read().strip().split("\n")[-1])
    full = json.load(open(os.path.join(root, "profiles", "r06_bench_default_full.json")))
    rows = list(csv.DictReader(open(os.path.join(root, "profiles", "r06_rcnet_b8_256x512_bf16_kernel_stats.csv"))))
    assert line["metric"] == "train imgs/sec (RC-Net+SML, 256x512)" and line["config"]["global_batch"] == 16
    checked = 0
    for fam in full["roofline_families"][:4]:
        names = set(fam["instantiation_names"])      # the instantiations whose work the engine tallied (bench.py build_rooflines)
        sel = [r for r in rows if kernel_key(r["Name"]) in names]
        assert {kernel_key(r["Name"]) for r in sel} == names, ("the library names an instantiation the trace does not have", names - {kernel_key(r["Name"]) for r in sel})
        calls, ns = sum(int(r["Calls"]) for r in sel), sum(float(r["TotalDurationNs"]) for r in sel)
        assert calls > 0, fam["kernel"]
        avg_s = ns / calls * 1e-9
        work = fam["algorithmic_flops_per_launch"] if fam["bound"] == "mfma" else fam["algorithmic_bytes_per_launch"]
        frac = work / avg_s / (1e12 if fam["bound"] == "mfma" else 1e9) / fam["peak"]
        assert abs(frac - fam["frac"]) <= 0.02 * fam["frac"], (fam["kernel"], frac, fam["frac"])
        checked += 1
    assert checked == 4 and abs(line["roofline"]["frac"] - full["roofline_families"][0]["frac"]) < 1e-4
