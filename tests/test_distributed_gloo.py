"""world_size-2 check (gloo, CPU) of the data-parallel gradient exchange: flat-arena slices are summed across ranks
and the 1/world average is folded into the optimizer's grad_scale."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from riders_amd.optim import FlatAdam
    from riders_amd.parallel import GradientAllReducer
    torch.manual_seed(rank)  # different initial weights per rank on purpose
    params = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(33)), torch.nn.Parameter(torch.randn(4, 3, 3, 3))]
    opt = FlatAdam(params, lr=1e-3)
    red = GradientAllReducer(opt, bucket_bytes=64)  # several buckets
    red.broadcast_parameters(0)
    ref0 = torch.cat([p.detach().reshape(-1) for p in params]).clone()
    for p in params:
        opt._grad_view(p).copy_(torch.full_like(p, float(rank + 1)))
    red.reduce()
    ok = all(bool(torch.all(opt._grad_view(p) == 3.0)) for p in params)  # 1 + 2
    dist.barrier()
    dist.destroy_process_group()
    torch.save((rank, ok, opt.grad_scale, ref0, len(red.buckets)), os.path.join(outdir, "r%d.pt" % rank))


def test_arena_allreduce_world2():
    import tempfile
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_worker, args=(r, 2, port, outdir)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=180)
            assert p.exitcode == 0, "worker exit code %s" % p.exitcode
        res = [torch.load(os.path.join(outdir, "r%d.pt" % r)) for r in range(2)]
    assert res[0][1] and res[1][1], "gradient sum wrong"
    assert res[0][2] == 0.5 and res[1][2] == 0.5
    assert torch.equal(res[0][3], res[1][3]), "parameters not broadcast from rank 0"
    assert res[0][4] > 1


def _ddp_worker(rank, world, port, outdir, emu_lib, transport="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from riders_amd import _lib, engine, rcnet_main
    from riders_amd.optim import FlatAdam
    from riders_amd.parallel import GradientAllReducer, RcclComm
    from tests.test_host_logic import _TwoStage, _toy_batch, _toy_loss
    _lib._install_for_tests(emu_lib)      # TEST INFRA: host build of the kernel sources (no GPU in this container)
    engine.set_compute_dtype("fp32")
    torch.manual_seed(100 + rank)          # different initial weights per rank on purpose: broadcast must fix it
    model = _TwoStage(); model.train()
    opt = FlatAdam(list(model.parameters()), lr=1e-2)
    # transport "c_abi" / "c_abi_rs_ag": the gradients travel through rd_comm_* (csrc/rd_comm.cpp; here its emulator build's shared-memory
    # transport, one process per rank as on the GPUs) -- torch.distributed (gloo) only carries the 128-byte rendezvous id and the vote
    comm = RcclComm(rank, world) if transport != "gloo" else None
    red = GradientAllReducer(opt, stages={"head_done": list(model.head.parameters()) + list(model.out.parameters())}, comm=comm,
                             mode="rs_ag" if transport.endswith("rs_ag") else "all_reduce", bucket_bytes=(1 << 10) if comm is not None else (32 << 20))
    red.broadcast_parameters(0)
    init = opt.flat_param.clone()
    grads, order = [], []
    for step in range(2):
        x, label, valid = _toy_batch(10 * step + rank)      # this rank's shard
        rcnet_main.staged_gradients(lambda: _toy_loss(model, x, label, valid), opt)
        order.append([t for t, _, _ in red.log])      # what was started from the stage marks, before reduce()
        red.reduce()
        red.log = []
        grads.append(opt.flat_grad.clone() * opt.grad_scale)
        opt.step()
    if comm is not None:
        assert comm.pending() == 0
        comm.close()
    dist.barrier()
    dist.destroy_process_group()
    torch.save(dict(rank=rank, init=init, grads=grads, final=opt.flat_param.clone(), order=order), os.path.join(outdir, "r%d.pt" % rank))


import pytest  # noqa: E402


@pytest.mark.parametrize("transport", ["gloo", "c_abi", "c_abi_rs_ag"])
def test_two_training_steps_world2_match_per_shard_mean(emu_lib_path, transport):
    """SURVEY 8(e) semantics: every rank runs the full step on its own shard (per-rank BatchNorm statistics, per-rank loss
    normaliser), the stage-bucketed all-reduce averages the gradients, Adam applies the average.  Two steps on two ranks must equal a
    single process that evaluates both shards on the same weights and applies the mean gradient; the head bucket is issued at its
    stage mark, i.e. before the body's backward has run.  Transports: torch.distributed's gloo all-reduce, and the library's own
    communicator behind the C ABI (parallel.RcclComm -> rd_comm_init / rd_allreduce_bucket / rd_comm_broadcast / rd_comm_join) in both bucket
    modes -- on this GPU-less container through the emulator build's shared-memory transport, two PROCESSES as on the GPUs."""
    import tempfile
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, outdir, emu_lib_path, transport)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=600)
            assert p.exitcode == 0, "worker exit code %s" % p.exitcode
        res = [torch.load(os.path.join(outdir, "r%d.pt" % r)) for r in range(2)]
    assert torch.equal(res[0]["init"], res[1]["init"]), "parameters not broadcast from rank 0"
    assert torch.equal(res[0]["final"], res[1]["final"]), "ranks diverged"
    assert all(o and set(o) == {"head_done"} for o in res[0]["order"]), res[0]["order"]     # bucket(s) started from the stage mark, before reduce()
    for s in range(2):
        assert torch.equal(res[0]["grads"][s], res[1]["grads"][s])
    # single-process reference: same initial weights, per-shard gradients on identical weights, mean, Adam
    from riders_amd import _lib, engine, rcnet_main
    from riders_amd.optim import FlatAdam
    from tests.test_host_logic import _TwoStage, _toy_batch, _toy_loss
    _lib._install_for_tests(emu_lib_path)
    try:
        engine.set_compute_dtype("fp32")
        torch.manual_seed(100)
        model = _TwoStage(); model.train()
        opt = FlatAdam(list(model.parameters()), lr=1e-2)
        assert torch.equal(opt.flat_param, res[0]["init"])
        for step in range(2):
            per = []
            for rank in range(2):
                x, label, valid = _toy_batch(10 * step + rank)
                rcnet_main.staged_gradients(lambda: _toy_loss(model, x, label, valid), opt)
                per.append(opt.flat_grad.clone())
            mean = (per[0] + per[1]) * 0.5
            err = float((mean - res[0]["grads"][step]).abs().max() / mean.abs().max())
            assert err < 1e-6, "step %d: all-reduced gradient differs from the per-shard mean (%.2e)" % (step, err)
            opt.flat_grad.copy_(mean)
            opt.step()
        err = float((opt.flat_param - res[0]["final"]).abs().max() / opt.flat_param.abs().max())
        assert err < 1e-6, "parameters after two data-parallel steps differ from the single-process reference (%.2e)" % err
    finally:
        engine.clear_caches()
        engine.set_param_grad_allocator(None)
        _lib._uninstall_for_tests()


def _sml_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contextlib
    import io
    from riders_amd import sml_main
    from riders_amd.optim import FlatAdam
    from riders_amd.parallel import GradientAllReducer, sml_stages
    torch.manual_seed(200 + rank)          # different initial weights per rank on purpose
    with contextlib.redirect_stdout(io.StringIO()):
        model = sml_main.build_model(torch.device("cpu"))      # the real MidasNet_small_videpth: its parameter list and stage split are what is tested
    opt = FlatAdam(list(model.parameters()), lr=1e-4)
    red = GradientAllReducer(opt, stages=sml_stages(model), bucket_bytes=4 << 20)
    red.broadcast_parameters(0)
    init = opt.flat_param.clone()
    # the backward of a step, as the engine drives it: gradients land in the arena, the stage marks fire in backward order, then reduce().
    # (The kernels themselves run in the single-process emulator / GPU tests; a whole SML step on the host emulator takes ~20 minutes.)
    g = torch.Generator().manual_seed(300 + rank)
    local = torch.randn(opt.numel, generator=g)
    opt.flat_grad.copy_(local)
    red.on_stage("scratch_done")
    red.on_stage("layer4_done")
    order = [t for t, _, _ in red.log]
    red.reduce()
    log = list(red.log)
    summed = opt.flat_grad.clone()
    dist.barrier()
    dist.destroy_process_group()
    torch.save(dict(init=init, order=order, log=log, local=local, summed=summed, scale=opt.grad_scale, numel=opt.numel), os.path.join(outdir, "r%d.pt" % rank))


def test_sml_arena_world2_stage_buckets():
    """The Scale Map Learner's gradient exchange on two ranks: parameters are broadcast, the scratch-decoder and layer4 buckets
    (parallel.sml_stages on the real model) are issued from their stage marks in backward order before reduce(), every arena element is
    reduced exactly once, and the reduced arena is the sum of the two ranks' local gradients."""
    import tempfile
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_sml_worker, args=(r, 2, port, outdir)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
            assert p.exitcode == 0, "worker exit code %s" % p.exitcode
        res = [torch.load(os.path.join(outdir, "r%d.pt" % r)) for r in range(2)]
    assert torch.equal(res[0]["init"], res[1]["init"]), "parameters not broadcast from rank 0"
    assert sorted(set(res[0]["order"]), key=res[0]["order"].index) == ["scratch_done", "layer4_done"], res[0]["order"]
    assert res[0]["scale"] == 0.5
    cover = torch.zeros(res[0]["numel"], dtype=torch.int32)
    for _, s, e in res[0]["log"]:
        cover[s:e] += 1
    assert bool((cover == 1).all()), "arena elements reduced %d..%d times" % (int(cover.min()), int(cover.max()))
    tot = res[0]["local"] + res[1]["local"]
    for r in range(2):
        assert torch.equal(res[r]["summed"], tot), "rank %d: all-reduced arena differs from the sum of the local gradients" % r


def _rendezvous_worker(rank, world, port, outdir, emu_lib):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from riders_amd import _lib
    from riders_amd.parallel import RcclComm
    _lib._install_for_tests(emu_lib)
    lib = _lib.load()
    real = lib.rd_comm_unique_id
    out = {}
    # (1) rank 0 cannot draw an id: every rank raises, nobody is left in the broadcast
    if rank == 0:
        lib.rd_comm_unique_id = lambda buf: -2
    try:
        RcclComm(rank, world)
        out["id_failure"] = "no error"
    except RuntimeError as ex:
        out["id_failure"] = str(ex)
    lib.rd_comm_unique_id = real
    # (2) rank 1 cannot bind the transport: every rank raises BEFORE rd_comm_init (which would block the others)
    avail = lib.rd_comm_available
    if rank == 1:
        lib.rd_comm_available = lambda: -2
    try:
        RcclComm(rank, world)
        out["bind_failure"] = "no error"
    except RuntimeError as ex:
        out["bind_failure"] = str(ex)
    lib.rd_comm_available = avail
    # (3) and afterwards the same ranks still build a working communicator
    c = RcclComm(rank, world)
    buf = torch.full((5,), float(rank + 1))
    c.all_reduce(buf)
    c.join(buf)
    out["sum"] = buf.tolist()
    c.close()
    dist.barrier()
    dist.destroy_process_group()
    torch.save(out, os.path.join(outdir, "r%d.pt" % rank))


def test_comm_rendezvous_fails_on_every_rank_together(emu_lib_path):
    """a rank that fails before the rendezvous must not leave its peers blocked -- rank 0 failing to draw the id, or any rank
    failing to bind the transport, raises on EVERY rank (after the id exchange / the vote), and a later attempt still works."""
    import tempfile
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_rendezvous_worker, args=(r, 2, port, outdir, emu_lib_path)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
            assert p.exitcode == 0, "worker exit code %s" % p.exitcode
        res = [torch.load(os.path.join(outdir, "r%d.pt" % r)) for r in range(2)]
    for r in range(2):
        assert "not every rank" in res[r]["id_failure"], res[r]
        assert "not every rank" in res[r]["bind_failure"], res[r]
        assert res[r]["sum"] == [3.0] * 5, res[r]
