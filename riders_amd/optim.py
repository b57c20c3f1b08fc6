"""Flat-arena fused Adam: every parameter (and its gradient and both moments) lives in ONE contiguous fp32
buffer, so an optimizer step is a single rd_adam_step launch and data-parallel gradient exchange can all-reduce
arena slices in place (no bucket copies).

Reference: torch.optim.Adam(parameters, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) at
RCNet/rcnet_main.py:233-238 and train_zju.py:205-211; same update arithmetic, same `param_groups[0]['lr']`
handle for the learning-rate schedule (rcnet_main.py:246-252, train_zju.py:231-237).
"""
import ctypes

import torch

from . import _lib, engine


class FlatAdam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = [p for p in params]
        if len(params) == 0:
            raise ValueError("FlatAdam got an empty parameter list")
        dev = params[0].device
        for p in params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatAdam needs fp32 parameters on one device")
        self.params = params
        self.param_groups = [dict(params=params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        # 16-byte aligned slots so the vectorised kernel and RCCL see aligned slices
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.numel = n
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._gviews = {}
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                view = self.flat_param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                self._gviews[id(p)] = self.flat_grad[o:o + p.numel()].view(p.shape)
        self.step_count = 0
        self.grad_scale = 1.0
        engine.set_param_grad_allocator(self._grad_view)

    def _grad_view(self, p):
        return self._gviews.get(id(p))

    def zero_grad(self, set_to_none=True):
        """Gradients are (re)written by the backward kernels into arena views; parameters that received no gradient
        keep a zero slot (Adam then leaves them unchanged, as torch does for grad=None with weight_decay 0)."""
        for p in self.params:
            p.grad = None

    def _gather_foreign_grads(self):
        # gradients produced outside the arena (user-assigned tensors): copy in with a HIP kernel
        for p in self.params:
            g = p.grad
            if g is None:
                continue
            view = self._gviews[id(p)]
            if g.data_ptr() != view.data_ptr():
                gc = g if g.is_contiguous() else g.contiguous()
                engine._chk(engine.L().rd_cast(engine._p(gc), engine._p(view), gc.numel(), engine.rd_of(gc), 0, 1.0,
                                               engine._stream(gc)), "rd_cast")

    def step(self):
        self._gather_foreign_grads()
        self.step_count += 1
        g = self.param_groups[0]
        fp = self.flat_param
        rc = engine.L().rd_adam_step(engine._p(fp), engine._p(self.flat_grad), engine._p(self.exp_avg), engine._p(self.exp_avg_sq),
                                     self.numel, ctypes.c_float(g['lr']), ctypes.c_float(g['betas'][0]),
                                     ctypes.c_float(g['betas'][1]), ctypes.c_float(g['eps']), ctypes.c_float(g['weight_decay']),
                                     self.step_count, ctypes.c_float(self.grad_scale), engine._stream(fp))
        engine._chk(rc, "rd_adam_step")
        engine.refresh_packed()   # one launch re-packs every cached MFMA operand of the rewritten parameters

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, param_groups=[
            {k: v for k, v in self.param_groups[0].items() if k != 'params'}])

    def load_state_dict(self, sd):
        self.step_count = int(sd['step'])
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.param_groups[0].update(sd['param_groups'][0])
