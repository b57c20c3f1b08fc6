"""Data parallelism: one process per GPU, gradients summed over RCCL (torch.distributed backend "nccl" on ROCm).

The reference only has single-process torch.nn.DataParallel (RCNet/rcnet_model.py:259-265, val_zju.py:341) which
cannot shard the per-image box list; here each rank runs the full step on its shard of the batch (per-rank
BatchNorm statistics, as DataParallel has) and the only exchange is a sum-all-reduce of the flat gradient arena,
issued in bucket order on a side stream as soon as a region's backward has produced its slice.
"""
import torch
import torch.distributed as dist


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
    """All-reduce the FlatAdam gradient arena across ranks (sum; the 1/world average is folded into Adam's
    grad_scale).  `buckets` slices are reduced asynchronously on torch.distributed's communication stream, so the
    reduction of early buckets overlaps the remaining backward work queued on the compute stream."""

    def __init__(self, optimizer, bucket_bytes=32 << 20, process_group=None):
        self.opt = optimizer
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        n = optimizer.numel
        per = max(1, bucket_bytes // 4)
        self.buckets = [(s, min(n, s + per)) for s in range(0, n, per)]
        optimizer.grad_scale = 1.0 / self.world
        self._handles = []

    def broadcast_parameters(self, src=0):
        if self.world > 1:
            dist.broadcast(self.opt.flat_param, src, group=self.group)

    def reduce(self):
        """Call after backward(); returns when every bucket's all-reduce has been enqueued and waited on."""
        if self.world == 1:
            return
        # buckets hold decoder grads last in arena order; reduce back-to-front = reverse execution order
        for s, e in reversed(self.buckets):
            self._handles.append(dist.all_reduce(self.opt.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in self._handles:
            h.wait()
        self._handles = []
