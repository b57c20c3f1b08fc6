"""Flat-arena fused Adam: every parameter (and its gradient and both moments) lives in ONE contiguous fp32
buffer, so an optimizer step is a single rd_adam_step launch and data-parallel gradient exchange can all-reduce
arena slices in place (no bucket copies).

Reference: torch.optim.Adam(parameters, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) at
RCNet/rcnet_main.py:233-238 and train_zju.py:205-211; same update arithmetic, same `param_groups[0]['lr']`
handle for the learning-rate schedule (rcnet_main.py:246-252, train_zju.py:231-237), and the same
`state_dict()` layout ({'state': {index: {step, exp_avg, exp_avg_sq}}, 'param_groups': [...]}) so the
`radarnet_optimizer_state_dict` entry of a checkpoint (RCNet/rcnet_model.py:224-257) is exchangeable with the
reference's torch.optim.Adam in both directions.

torch skips a parameter whose `.grad` is None (no moment update, no step increment).  The same holds here: a slot
that no backward kernel wrote since `zero_grad()` is left out of the step.  A slot that has NEVER been written
holds zero gradient and zero moments, for which the update is exactly zero, so the common case -- the reference's
never-used `projection` convolutions (utils/net_utils.py:300-307) -- still runs as one launch over the arena.
"""
import ctypes

import torch

from . import engine

_GROUP_DEFAULTS = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                       decoupled_weight_decay=False)


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
        self._index = {}
        self._owner = frozenset(id(p) for p in params)      # engine.refresh_packed: whose cached operands this optimizer's step re-packs
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(params, self.offsets)):
                view = self.flat_param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                self._gviews[id(p)] = self.flat_grad[o:o + p.numel()].view(p.shape)
                self._index[id(p)] = i
        self.steps = [0] * len(params)          # per-parameter step count (torch keeps `step` per parameter)
        self._touched = [False] * len(params)   # slot written by a backward kernel since zero_grad()
        self._ever = [False] * len(params)      # slot has received a gradient at least once (torch: has optimizer state)
        self.grad_scale = 1.0      # 1 / world size, set by parallel.GradientAllReducer (the all-reduce sums)
        self.loss_scale = 1.0      # static loss scale of the fp16 mode: gradients arrive multiplied by it and are divided here
        self._overflow = None      # device int32[2] (flag, skipped steps) of the guarded fp16 step, allocated on first use
        self._skips_reconciled = 0  # skipped steps already taken back out of the host step counters (reconcile_skipped)
        engine.set_param_grad_allocator(self._grad_view)

    # kept for callers that read the global step (all parameters that train share it)
    @property
    def step_count(self):
        return max(self.steps) if self.steps else 0

    def _grad_view(self, p):
        i = self._index.get(id(p))
        if i is None:
            return None
        self._touched[i] = True
        return self._gviews[id(p)]

    def mark_touched(self, indices):
        """Declare these slots written since zero_grad().  A hipGraph REPLAY of the backward writes the arena without going through the
        gradient allocator (which is what sets the flags in an eager backward): rcnet_main.GraphedStep records the slots its capture pass
        touched and re-marks them before every step(), so `optimizer.zero_grad()` between replays -- the reference loop's habit -- cannot
        turn the optimizer step into a silent no-op."""
        for i in indices:
            self._touched[i] = True

    def touched_indices(self):
        return [i for i, t in enumerate(self._touched) if t]

    def skipped_steps(self):
        """fp16 mode: optimizer steps dropped because a scaled gradient was not finite (synchronises).  The host-side step counters (bias
        correction) are advanced before the device decides to skip, so after k skipped steps they run k ahead of the moments until
        `reconcile_skipped()` (called by state_dict()) takes them back -- torch's GradScaler likewise does not advance `step` on a skip."""
        return 0 if self._overflow is None else int(self._overflow[1].item())

    def reconcile_skipped(self, warn_after=8):
        """Subtract the device-side skip count from the host step counters (synchronises; the loss scale is static, so a persistent overflow
        skips EVERY step: more than `warn_after` skips since the last call raise a warning naming the remedy).  -> skips taken back."""
        k = self.skipped_steps()
        new = k - self._skips_reconciled
        if new > 0:
            self.steps = [max(0, s - new) if e else s for s, e in zip(self.steps, self._ever)]
            self._skips_reconciled = k
            if new > warn_after:
                import warnings
                warnings.warn("FlatAdam: %d optimizer steps were skipped because the scaled fp16 gradient was not finite; lower loss_scale "
                              "(now %g)" % (new, self.loss_scale))
        return max(new, 0)

    def slot_range(self, params):
        """-> sorted, merged [(start, end)] arena element ranges covering `params` (used to bucket the gradient all-reduce)."""
        spans = sorted((self.offsets[self._index[id(p)]], self.offsets[self._index[id(p)]] + (p.numel() + 3) // 4 * 4) for p in params)
        out = []
        for s, e in spans:
            if out and s <= out[-1][1]:
                out[-1] = (out[-1][0], max(out[-1][1], e))
            else:
                out.append((s, e))
        return out

    def zero_grad(self, set_to_none=True):
        """Gradients are (re)written by the backward kernels into arena views; a slot nobody writes before step() is skipped by it."""
        for p in self.params:
            p.grad = None
        self._touched = [False] * len(self.params)

    def _gather_foreign_grads(self):
        # gradients produced outside the arena (user-assigned tensors): copy in with a HIP kernel
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                continue
            view = self._gviews[id(p)]
            self._touched[i] = True
            if g.data_ptr() != view.data_ptr():
                gc = g if g.is_contiguous() else g.contiguous()
                engine._chk(engine.L().rd_cast(engine._p(gc), engine._p(view), gc.numel(), engine.rd_of(gc), 0, 1.0,
                                               engine._stream(gc)), "rd_cast")

    def _launch(self, start, end, step):
        g = self.param_groups[0]
        sl = slice(start, end)
        args = (engine._p(self.flat_param[sl]), engine._p(self.flat_grad[sl]), engine._p(self.exp_avg[sl]), engine._p(self.exp_avg_sq[sl]), end - start,
                ctypes.c_float(g['lr']), ctypes.c_float(g['betas'][0]), ctypes.c_float(g['betas'][1]), ctypes.c_float(g['eps']),
                ctypes.c_float(g['weight_decay']), step, ctypes.c_float(self.grad_scale / self.loss_scale))
        st = engine._stream(self.flat_param)
        if self._overflow is not None:
            rc = engine._tb("optimizer", 28 * (end - start), lambda: engine.L().rd_adam_step_guarded(*args, engine._p(self._overflow), st), "adam")
        else:
            rc = engine._tb("optimizer", 28 * (end - start), lambda: engine.L().rd_adam_step(*args, st), "adam")
        engine._chk(rc, "rd_adam_step")

    def step(self):
        self._gather_foreign_grads()
        n = len(self.params)
        for i in range(n):
            if self._touched[i]:
                self.steps[i] += 1
                self._ever[i] = True
        live = [s for s, t in zip(self.steps, self._touched) if t]
        if not live:
            return
        guarded = self.loss_scale != 1.0     # fp16 mode: a non-finite scaled gradient skips the whole update (device-side, no host sync)
        if guarded:
            if self._overflow is None:
                self._overflow = torch.zeros(2, dtype=torch.int32, device=self.flat_param.device)
            engine._chk(engine.L().rd_grad_finite_check(engine._p(self.flat_grad), self.numel, engine._p(self._overflow),
                                                        engine._stream(self.flat_param)), "rd_grad_finite_check")
        # one launch when every slot is either written this step (all at the same step count) or has never been written (zero gradient and
        # zero moments: the update is exactly zero); otherwise one launch per run of consecutive written slots that share a step count
        uniform = len(set(live)) == 1 and all(t or not e for t, e in zip(self._touched, self._ever)) and self.param_groups[0]['weight_decay'] == 0
        if uniform:
            self._launch(0, self.numel, live[0])
        else:
            i = 0
            while i < n:
                if not self._touched[i]:
                    i += 1
                    continue
                j = i
                while j + 1 < n and self._touched[j + 1] and self.steps[j + 1] == self.steps[i]:
                    j += 1
                end = self.offsets[j + 1] if j + 1 < n else self.numel
                self._launch(self.offsets[i], end, self.steps[i])
                i = j + 1
        if guarded:
            engine._chk(engine.L().rd_adam_skip_count(engine._p(self._overflow), engine._stream(self.flat_param)), "rd_adam_skip_count")
        engine.refresh_packed(self._owner)   # one launch re-packs every cached MFMA operand of the rewritten parameters (this optimizer's only)

    # ------------------------------------------------------------------ checkpoint interchange with torch.optim.Adam
    def state_dict(self):
        """torch.optim.Adam's layout: per-parameter `step` / `exp_avg` / `exp_avg_sq` (copies sliced out of the arena) for every parameter
        that has received a gradient, and one param group listing parameter indices."""
        state = {}
        engine.check_roi_overflow()   # (the host waits here anyway) a checkpoint must not be written from NaN-poisoned moments
        self.reconcile_skipped()      # `step` = updates actually applied (fp16 mode: not the skipped ones)
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            if not self._ever[i]:
                continue
            state[i] = dict(step=torch.tensor(float(self.steps[i])),
                            exp_avg=self.exp_avg[o:o + p.numel()].view(p.shape).clone(),
                            exp_avg_sq=self.exp_avg_sq[o:o + p.numel()].view(p.shape).clone())
        g = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        for k, v in _GROUP_DEFAULTS.items():
            g.setdefault(k, v)
        g['params'] = list(range(len(self.params)))
        return dict(state=state, param_groups=[g])

    def load_state_dict(self, sd):
        """Accepts a torch.optim.Adam state_dict (the reference's `radarnet_optimizer_state_dict`), this class's own output, and the
        flat {'step','exp_avg','exp_avg_sq'} form written by round-1 checkpoints."""
        if 'state' not in sd:      # round-1 private layout
            self.steps = [int(sd['step'])] * len(self.params)
            self._ever = [True] * len(self.params)
            self.exp_avg.copy_(sd['exp_avg'])
            self.exp_avg_sq.copy_(sd['exp_avg_sq'])
            self.param_groups[0].update({k: v for k, v in sd['param_groups'][0].items() if k != 'params'})
            self._skips_reconciled = self.skipped_steps()
            return
        groups = sd['param_groups']
        order = [i for g in groups for i in g['params']]
        if len(order) != len(self.params):
            raise ValueError("loaded state dict has %d parameters, the optimizer has %d" % (len(order), len(self.params)))
        if any(bool(g.get('amsgrad', False)) or bool(g.get('maximize', False)) for g in groups):
            raise ValueError("amsgrad / maximize Adam states are not supported (the reference uses neither)")
        with torch.no_grad():
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            self.steps = [0] * len(self.params)
            self._ever = [False] * len(self.params)
            for pos, key in enumerate(order):
                st = sd['state'].get(key)
                if st is None:
                    continue
                p, o = self.params[pos], self.offsets[pos]
                if tuple(st['exp_avg'].shape) != tuple(p.shape):
                    raise ValueError("optimizer state %r has shape %s, parameter %d has %s" % (key, tuple(st['exp_avg'].shape), pos, tuple(p.shape)))
                self.exp_avg[o:o + p.numel()].view(p.shape).copy_(st['exp_avg'])
                self.exp_avg_sq[o:o + p.numel()].view(p.shape).copy_(st['exp_avg_sq'])
                self.steps[pos] = int(float(st['step']))
                self._ever[pos] = True
        # the loaded `step` values are updates actually applied: skips counted on the device so far belong to the state that was replaced
        self._skips_reconciled = self.skipped_steps()
        g0 = {k: v for k, v in groups[0].items() if k in ('lr', 'betas', 'eps', 'weight_decay')}
        if 'betas' in g0:
            g0['betas'] = tuple(g0['betas'])
        self.param_groups[0].update(g0)
