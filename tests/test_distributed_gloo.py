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
