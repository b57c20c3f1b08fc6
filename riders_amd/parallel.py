"""Data parallelism: one process per GPU, gradients summed over RCCL (torch.distributed backend "nccl" on ROCm).

The reference only has single-process torch.nn.DataParallel (RCNet/rcnet_model.py:259-265, val_zju.py:341) which
cannot shard the per-image box list; here each rank runs the full step on its shard of the batch (per-rank
BatchNorm statistics, as DataParallel has) and the only exchange is a sum-all-reduce of the flat gradient arena.

Overlap with backward: the models place `engine.stage_mark(tag)` boundaries in their forward (RC-Net: decoder |
transformer + point MLP | image encoder; SML: scratch decoder | backbone layer4 | rest).  When the backward passes a
mark, the gradients of everything after it are final; `GradientAllReducer.on_stage(tag)` then starts the asynchronous
all-reduce of that bucket (arena slices, in place) on torch.distributed's communication stream, which waits only for
the work queued so far, so the exchange runs while the earlier layers are still back-propagating on the compute stream.
`reduce()` starts whatever has not been started and waits for everything before the optimizer step.
Buckets are whole stages (7-9 MB for RC-Net): xGMI is point-to-point, a ring collective is per-link bound, so few large
messages beat many small ones.
"""
import ctypes

import torch
import torch.distributed as dist

from . import engine


class RcclComm(object):
    """The library's own RCCL communicator (include/riders_hip.h rd_comm_*: ncclCommInitRank on this process's device + a library-owned
    communication stream).  The 128-byte rendezvous id is drawn by rank 0 and handed to the other ranks through `exchange(bytes or None) ->
    bytes` -- by default a broadcast over an initialised torch.distributed group (any backend), which is used for this ONE host-side
    exchange only; with world == 1 nothing is exchanged.  The collectives themselves never pass through torch.distributed: they are stream
    operations of libriders_hip.so and can be captured into the step's hipGraph (rcnet_main.GraphedStep)."""

    def __init__(self, rank=None, world=None, exchange=None, agree=None):
        """exchange(bytes or None) -> bytes or None: hands rank 0's rendezvous id to every rank (default: torch.distributed broadcast);
        agree(bool) -> bool: True iff EVERY rank passed True (default: a MIN all-reduce on the host side of torch.distributed).
        The ranks fail TOGETHER: a rank that cannot bind RCCL, or a rank 0 that cannot draw an id, still takes part in both exchanges, so no
        peer is left waiting in a broadcast or inside ncclCommInitRank (which blocks until every rank has entered it)."""
        lib = engine.L()
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank, self.world = int(rank), int(world)
        self.handle = None
        idb = ctypes.create_string_buffer(128)
        err = None
        ok = lib.rd_comm_available() == 0
        if not ok:
            err = "rank %d: %s" % (self.rank, (lib.rd_last_error_string() or b"RCCL cannot be bound").decode(errors="replace"))
        if self.rank == 0 and ok:
            if lib.rd_comm_unique_id(idb) != 0:
                ok, err = False, "rank 0: %s" % (lib.rd_last_error_string() or b"rd_comm_unique_id failed").decode(errors="replace")
        if self.world > 1:
            if exchange is None or agree is None:
                if not dist.is_initialized():
                    raise RuntimeError("RcclComm: world > 1 needs an initialised torch.distributed group or `exchange` / `agree` callables for the rendezvous")
            if exchange is None:
                def exchange(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            if agree is None:
                def agree(flag):
                    votes = [None] * self.world
                    dist.all_gather_object(votes, bool(flag))
                    return all(votes)
            got = exchange((bytes(idb.raw) if ok else None) if self.rank == 0 else None)      # None = rank 0 could not draw an id
            if got is None:
                ok, err = False, err or "rank 0 could not create the rendezvous id"
            else:
                idb = ctypes.create_string_buffer(bytes(got), 128)
            if not agree(ok):
                raise RuntimeError("RcclComm: not every rank can create the communicator (%s)" % (err or "another rank failed"))
        elif not ok:
            raise RuntimeError("RcclComm: " + err)
        h = ctypes.c_void_p()
        engine._chk(lib.rd_comm_init(self.rank, self.world, idb, ctypes.byref(h)), "rd_comm_init")
        self.handle = h

    def all_reduce(self, buf, mode=0):
        """in-place fp32 sum of `buf` over the ranks on the communication stream, ordered behind the work queued on buf's current stream"""
        assert buf.dtype == torch.float32 and buf.is_contiguous()
        engine._chk(engine.L().rd_allreduce_bucket(self.handle, engine._p(buf), buf.numel(), int(mode), engine._stream(buf)), "rd_allreduce_bucket")

    def broadcast(self, buf, root=0):
        assert buf.dtype == torch.float32 and buf.is_contiguous()
        engine._chk(engine.L().rd_comm_broadcast(self.handle, engine._p(buf), buf.numel(), int(root), engine._stream(buf)), "rd_comm_broadcast")

    def join(self, like):
        """the current stream of `like`'s device waits for every collective issued since the last join (the host does not block)"""
        engine._chk(engine.L().rd_comm_join(self.handle, engine._stream(like)), "rd_comm_join")

    def pending(self):
        return int(engine.L().rd_comm_pending(self.handle))

    def close(self):
        if self.handle is not None:
            engine.L().rd_comm_destroy(self.handle)
            self.handle = None


class ModuleHolder(torch.nn.Module):
    """Keeps the `module.` state_dict prefix of the reference's DataParallel-wrapped checkpoints."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def _fwd(self, *a, **k):
        return self.module._fwd(*a, **k)


class GradientAllReducer(object):
    """All-reduce the FlatAdam gradient arena across ranks (sum; the 1/world average is folded into Adam's grad_scale).

    stages: optional {tag: iterable of parameters whose gradients are final when the backward passes stage mark `tag`}.
    Without stages (or for parameters in none) the arena is reduced by `reduce()` in `bucket_bytes` slices after backward."""

    def __init__(self, optimizer, bucket_bytes=32 << 20, process_group=None, stages=None, mode="all_reduce", comm=None):
        """comm: an RcclComm -- the collectives are the library's own (rd_allreduce_bucket on its communication stream, capturable: the step is
        ONE hipGraph); None: torch.distributed's (backend "nccl" = RCCL, or gloo in the CPU tests).
        mode: 'all_reduce' (RCCL picks the algorithm: a ring on xGMI) or 'rs_ag' = reduce_scatter_tensor + all_gather_into_tensor in
        place on each bucket (the two halves of a ring all-reduce issued separately: same bytes per link, but the gather half of bucket k
        can interleave with the scatter half of bucket k+1 on the communication stream; for A/B on an 8-GPU node, NCCL/RCCL backend only)."""
        if mode not in ("all_reduce", "rs_ag"):
            raise ValueError("GradientAllReducer: mode must be 'all_reduce' or 'rs_ag'")
        self.opt = optimizer
        self.mode = mode
        self.group = process_group
        self.comm = comm
        if comm is not None:
            self.world, self.collective = comm.world, True
        else:
            self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
            self.collective = dist.is_initialized()      # also with one rank: the same calls, a functional check of the path on a 1-GPU box
        self.per = max(4, (bucket_bytes // 4) & ~3)
        self.stage_ranges = {}
        covered = []
        for tag, params in (stages or {}).items():
            rs = optimizer.slot_range(list(params))
            self.stage_ranges[tag] = rs
            covered += rs
        # the rest of the arena (stages never cover everything: the first layers finish last), front to back
        covered.sort()
        self.rest, pos = [], 0
        for s, e in covered:
            if s < pos:
                raise ValueError("GradientAllReducer: stages overlap")
            if s > pos:
                self.rest.append((pos, s))
            pos = e
        if pos < optimizer.numel:
            self.rest.append((pos, optimizer.numel))
        self.buckets = [b for rs in list(self.stage_ranges.values()) + [self.rest] for b in self._split(rs)]
        optimizer.grad_scale = 1.0 / self.world
        self._handles = []
        self._done = set()
        self.log = []      # (tag, start, end) of every all-reduce of the current / last finished step: test / debugging aid
        self._step_over = False
        if self.stage_ranges:
            engine.add_stage_hook(self.on_stage)

    def _split(self, ranges):
        return [(s0, min(e, s0 + self.per)) for s, e in ranges for s0 in range(s, e, self.per)]

    def close(self):
        engine.remove_stage_hook(self.on_stage)

    def broadcast_parameters(self, src=0):
        if self.comm is not None:
            self.comm.broadcast(self.opt.flat_param, src)
            self.comm.join(self.opt.flat_param)
        elif self.collective:
            dist.broadcast(self.opt.flat_param, src, group=self.group)
        engine.refresh_packed()     # the packed MFMA operands cached by earlier forwards follow the new values (same buffers)

    def _issue(self, tag, ranges):
        if self._step_over:      # first bucket of a new step
            self.log, self._step_over = [], False
        for s, e in self._split(ranges):
            self.log.append((tag, s, e))
            if self.collective:
                self._handles.extend(self._sum(self.opt.flat_grad[s:e]))

    def _sum(self, buf):
        """Start the sum of one arena slice across the ranks -> the async work handles."""
        if self.comm is not None:      # library-owned communication stream; ordered by rd_comm_join, no handles
            self.comm.all_reduce(buf, 1 if self.mode == "rs_ag" else 0)
            return []
        w = self.world
        n = buf.numel() // w * w
        if self.mode != "rs_ag" or dist.get_backend(self.group) != "nccl" or n == 0:      # (gloo has no reduce_scatter_tensor)
            return [dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        rank = dist.get_rank(self.group)
        body, shard = buf[:n], buf[:n].view(w, -1)[rank]      # in place: rank r's shard is the r-th piece of the bucket (NCCL's in-place convention)
        hs = [dist.reduce_scatter_tensor(shard, body, op=dist.ReduceOp.SUM, group=self.group, async_op=True),
              dist.all_gather_into_tensor(body, shard, group=self.group, async_op=True)]      # same communication stream: ordered behind the scatter
        if n < buf.numel():     # the few elements that do not divide by the world size
            hs.append(dist.all_reduce(buf[n:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return hs

    def on_stage(self, tag):
        """Stage hook (called from the backward, or by a graph-replaying driver right after the stage's graph was enqueued)."""
        rs = self.stage_ranges.get(tag)
        if rs is None or tag in self._done:
            return
        self._done.add(tag)
        self._issue(tag, rs)

    def reduce(self):
        """Call after backward(): starts every bucket not started by a stage mark, then waits for all of them (the compute stream is
        made to wait for the communication stream; the host does not block)."""
        for tag, rs in self.stage_ranges.items():
            if tag not in self._done:
                self._issue(tag, rs)
        self._issue(None, self.rest)
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.comm is not None:
            self.comm.join(self.opt.flat_grad)
        self._done = set()
        self._step_over = True


def rcnet_stages(model):
    """Stage -> parameters for RCNetModel (marks are placed by RCNetModel.forward / RCNetEncoder._fwd)."""
    enc = model.encoder.module if hasattr(model.encoder, "module") else model.encoder
    dec = model.decoder.module if hasattr(model.decoder, "module") else model.decoder
    return {"decoder_done": list(dec.parameters()),
            "attention_done": list(enc.attention.parameters()) + list(enc.encoder_depth.parameters())}


def sml_stages(model):
    """Stage -> parameters for MidasNet_small_videpth (marks are placed by MidasNet_small_videpth._fwd)."""
    return {"scratch_done": list(model.scratch.parameters()),
            "layer4_done": list(model.pretrained.layer4.parameters())}
