"""Execution engine: a small explicit tape over the HIP kernels (no tracing compiler, no torch compute ops).

PyTorch provides device memory, streams and the outer autograd edge; everything between a region's inputs
and outputs is a hand-scheduled sequence of libriders_hip.so launches recorded on a `Tape`, whose backward is
replayed in reverse.  Activations are NHWC tensors of shape (N, H, W, C) (or (rows, C) for token matrices) in
the engine's activation dtype (fp32 or bf16); parameters and their gradients stay fp32 in the reference's
layouts so `state_dict`s are exchangeable with the reference.

A module's forward runs either inside an already-active tape (it is part of a larger fused region) or opens
its own region, which appears to torch.autograd as ONE node (`_Region`).
"""
import contextlib
import os
import ctypes
import weakref

import torch
import torch.distributed

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_RELU6, RD_BF16, RD_F16, RD_F32, ConvDesc

_TORCH_DT = {RD_F32: torch.float32, RD_BF16: torch.bfloat16, RD_F16: torch.float16}
_RD_DT = {torch.float32: RD_F32, torch.bfloat16: RD_BF16, torch.float16: RD_F16}

_state = {"dtype": RD_F32, "tape": None, "defer_wgrad": True, "fused_loftr": True, "deterministic_roi_pool": False, "roi_tile_min_blocks": 256,
          # A/B switches of the engine (set_switch; none of them is read from the environment by the engine itself -- tools pass RIDERS_OPTS,
          # parsed ONCE and validated by apply_opts() below)
          "fuse_upsample_bwd": True,       # the 2x2-summing data gradient of exact 2x up-sampling layers
          "loftr_cross_inplace": True,     # engine.CrossGrad
          "fuse_res_add": True,            # residual of a conv without BatchNorm / activation added in its epilogue
          "fuse_grad_add": True,           # second gradient contribution added in the data-gradient epilogue
          # Round 4: conv -> BatchNorm -> activation outputs stay virtual (LazyAct) and the consumer applies scale / shift / activation while
          # staging its input.  Level 0: never (the separate rd_affine_act pass everywhere); 1 (default): inside residual blocks (conv1 ->
          # conv2 staging, conv2 -> fused apply + add + activation: measured faster); 2: also the decoder chain (measured SLOWER on MI355X:
          # the 3x3 kernels are vector-issue bound and the staging arithmetic costs more than the HBM pass it replaces -- DESIGN.md)
          "lazy_bn": 1,
          "fusion_conv_first": True,   # SML FeatureFusionBlock: the 1x1 out_conv before the bilinear x2 (they commute; a quarter of the pixels)
          "up2_dgrad": True,       # ... and its data gradient at source resolution from the space-to-depth view of dy (rd_conv_desc.in_s2d)
          "up2_on_source": True,   # forward of an exact-2x UpConv2d as a 3x3 convolution of the SOURCE with per-parity-class pre-summed weights (rd_conv_desc.out_d2s)
          "bn_head": True,    # conv -> BatchNorm -> act -> one-channel 3x3 output convolution: the fused decoder-head kernels (rd_bn_head_*)
          # (round 4 measured two concurrency experiments SLOWER on MI355X / ROCm 7 and round 5 removed them from the product: convolution weight
          # gradients on a second stream -- RC-Net 1005 -> 952 img/s, 38 fork / join edges per step -- and the skip features' RoI poolings next to
          # the transformer -- 1059 -> 1049; DESIGN.md section 3 "Round 4")
          "roi_u8": True,                  # compact (one byte) RoI-pool arg-max; False: int32 indices
          # RoI-pool backward: auto = pixel-owner gather for launches of >= roi_tile_min_blocks workgroups, fp32 L2 atomics below;
          # gather / tile / atomic force one form
          "roi_bwd": "auto",
          "bn_recompute": True,            # the BatchNorm backward that does not read z
          # its reduce pass inside the data-gradient epilogue that produces dz (rd_conv_fusion.bn_y; register-fed 3x3 and implicit-GEMM kernels).
          # Built and MEASURED in round 5: 11 of RC-Net's 27 reduce launches go (bn_backward + finalize -0.13 ms per
          # step), the data gradients that carry the sums get 0.20 ms slower (one more tensor read + ~10 vector instructions per stored
          # element in epilogues that are already issue bound): 1021.8 -> 1007.1 img/s on one box, alternating.  Off by default.
          "bn_bwd_fused": False,
          "dw_fused_stats": True,          # depthwise convolution with the BatchNorm statistics in its epilogue; False: separate rd_bn_stats pass
          "pack_vec": True,                # the batched weight re-pack in 16-byte units with coalesced reads (round 6); False: the element-wise form
          "pack_map": True}                # ... launched with a block map (the blocks the operands need); False: 256 blocks per operand

_SWITCHES = {"fuse_upsample_bwd": bool, "loftr_cross_inplace": bool, "fuse_res_add": bool, "fuse_grad_add": bool, "lazy_bn": (0, 2), "bn_head": bool, "up2_on_source": bool, "up2_dgrad": bool, "fusion_conv_first": bool, "roi_u8": bool,
             "roi_bwd": ("auto", "gather", "tile", "atomic"), "bn_recompute": bool, "bn_bwd_fused": bool, "dw_fused_stats": bool, "pack_vec": bool, "pack_map": bool, "roi_tile_min_blocks": (0, 1 << 30),
             "defer_wgrad": bool, "fused_loftr": bool, "deterministic_roi_pool": bool}


def set_switch(name, value):
    """One validated entry point for the engine's A/B switches (tests, tools): unknown names and out-of-range values raise."""
    kind = _SWITCHES.get(name)
    if kind is None:
        raise KeyError("riders_amd.engine: unknown switch %r (known: %s)" % (name, ", ".join(sorted(_SWITCHES))))
    if kind is bool:
        value = bool(int(value)) if not isinstance(value, bool) else value
    elif isinstance(kind[0], int):
        value = int(value)
        if not kind[0] <= value <= kind[1]:
            raise ValueError("riders_amd.engine: switch %s = %r outside [%d, %d]" % (name, value, kind[0], kind[1]))
    elif value not in kind:
        raise ValueError("riders_amd.engine: switch %s = %r not one of %r" % (name, value, kind))
    _state[name] = value


def set_option(name, value):
    """Kernel-routing option of the library (include/riders_hip.h rd_set_option: "frag_v128", "conv3x3_min_blocks", ...); None clears it."""
    lib = L()
    if value is None:
        _chk(lib.rd_clear_option(name.encode()), "rd_clear_option")
    else:
        _chk(lib.rd_set_option(name.encode(), int(value)), "rd_set_option")


def apply_opts(spec):
    """"name=value,name=value": engine switches (set_switch) and, prefixed `rd.`, library routing options (set_option).  The ONE place a
    string from outside (tools: the RIDERS_OPTS environment variable, read by bench.py / tools/bench_*.py, never by the package) reaches them."""
    for kv in (spec or "").split(","):
        kv = kv.strip()
        if not kv:
            continue
        k, _, v = kv.partition("=")
        if k.startswith("rd."):
            set_option(k[3:], None if v == "" else int(v))
        else:
            set_switch(k, v if k == "roi_bwd" else int(v))


# how often a virtual activation was consumed in place / had to be written after all (tests assert that the fused routes are taken)
lazy_counts = {"fwd_fused": 0, "wgrad_fused": 0, "add_fused": 0, "materialized": 0, "bn_bwd_fused": 0, "head_fused": 0, "head_unfused_bwd": 0, "up2_fwd": 0, "up2_dgrad": 0}


def head_route(C):
    """True when a conv -> BatchNorm -> act layer with C output channels whose only consumer is a one-channel 3x3 output convolution should
    hand that consumer a LazyAct (conv_block then takes the fused decoder-head kernels, rd_bn_head_*)."""
    return bool(_state["bn_head"] and _state["lazy_bn"] >= 1 and _state["bn_recompute"] and C == 16)


def switch(name):
    """current value of an engine switch (set_switch)"""
    return _state[name]


def set_lazy_bn(level):
    """Consumer-side BatchNorm apply (LazyAct): 0 off, 1 residual blocks only (default), 2 every conv -> BatchNorm -> conv chain (True = 2).
    Results are bit-identical at every level (tests compare them)."""
    _state["lazy_bn"] = 2 if level is True else int(level)


def set_roi_tile_min_blocks(n):
    """Smallest launch (in workgroups) for which the RoI-pool backward uses a tiled kernel (pixel-owner gather, or LDS tile-accumulate when
    that is preferred); smaller maps use global atomics.  n = 0 additionally prefers the LDS tile-accumulate kernel (tests)."""
    _state["roi_tile_min_blocks"] = int(n)


def set_deterministic_roi_pool(flag):
    """RoI-pool backward through the pixel-owner gather kernel at EVERY map size (bit-reproducible: every gradient element written once, in
    a fixed order).  By default the large maps already use it (round 2: 0.12 / 0.10 / 0.09 ms for RC-Net's three large levels against
    0.26 / 0.19 / 0.16 ms for the LDS tile-accumulate kernel); the small ones go through fp32 L2 atomics, which is faster there."""
    _state["deterministic_roi_pool"] = bool(flag)



def set_fused_loftr(flag):
    """Run eligible LoFTR encoder layers (d_model 128, 8 heads, <= 32 tokens) through the fused per-ROI kernels (default on)."""
    _state["fused_loftr"] = bool(flag)


def fused_loftr():
    return _state["fused_loftr"]



def set_defer_wgrad(flag):
    """Group the weight gradients of 1x1 / linear layers into one launch at the end of each tape's backward (default on)."""
    _state["defer_wgrad"] = bool(flag)



def set_compute_dtype(dt):
    """Activation storage dtype of subsequently executed regions: 'fp32', 'bf16' or 'fp16' (accumulation is always fp32).
    fp16 (BASELINE.json configs[4]) has a narrow exponent: train with a static loss scale (rcnet_main / sml_main `loss_scale`, folded back
    into Adam's grad_scale) -- the loss is a mean over millions of pixels and its raw gradients underflow fp16."""
    _state["dtype"] = {"fp32": RD_F32, "bf16": RD_BF16, "fp16": RD_F16, RD_F32: RD_F32, RD_BF16: RD_BF16, RD_F16: RD_F16}[dt]


def compute_dtype():
    return _state["dtype"]


def act_dtype():
    return _TORCH_DT[_state["dtype"]]


def _stream(t):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    if not _lib.ALLOW_HOST_POINTERS:
        raise RuntimeError("riders_amd kernels need ROCm device tensors (got %s); there is no CPU path" % t.device)
    return ctypes.c_void_p(0)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(rc, what):
    if rc != 0:
        _lib.check(rc, what)


def L():
    return _lib.load()


L_ = L   # alias for functions whose signature uses L as a token count


def rd_of(t):
    return _RD_DT[t.dtype]


# ------------------------------------------------------------------------------------- kernel timing hook
class KernelTimer(object):
    """Optional per-launch HIP-event timing of the GEMM-class kernels (used by bench.py for the roofline line).
    Events are recorded on the stream the kernels are launched on (torch's current stream).

    repeat > 1: a launch declared idempotent (`idem`: convolution forward / data gradient, weight-gradient slabs -- pure functions of
    their inputs) is issued `repeat` times back to back between ONE pair of events and its duration is the elapsed time / repeat: the
    event pair costs a few microseconds per launch, which rocprofv3's kernel durations do not contain (round 2: 27 % over on 50-us
    launches).  Everything else is issued once."""

    def __init__(self, repeat=1):
        self.records = {}  # kind -> [(start_event, end_event, algorithmic_flops, desc, bytes, kernel name, launches between the events)]
        self.repeat = max(1, int(repeat))

    def run(self, kind, flops, fn, desc=None, nbytes=0.0, kernel=None, idem=False):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = self.repeat if idem else 1
        s.record()
        rc = fn()
        for _ in range(n - 1):
            fn()
        e.record()
        self.records.setdefault(kind, []).append((s, e, flops, desc, nbytes, kernel, n))
        return rc

    def detail(self):
        """-> {(kind, desc): (launches, total ms, total flops, total algorithmic bytes)} aggregated over identical launch shapes."""
        out = {}
        for kind, recs in self.records.items():
            for s, e, f, d, b, _k, r in recs:
                k = (kind, d)
                n, ms, fl, by = out.get(k, (0, 0.0, 0.0, 0.0))
                out[k] = (n + 1, ms + s.elapsed_time(e) / r, fl + f, by + b)
        return out

    def by_kernel(self):
        """-> {kernel name: dict(launches, ms, flops, bytes, shapes={desc: [launches, ms, flops, bytes]})} over the launches that carry a name."""
        out = {}
        for kind, recs in self.records.items():
            for s, e, f, d, b, k, r in recs:
                if not k:
                    continue
                o = out.setdefault(k, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, shapes={}, kind=kind))
                ms = s.elapsed_time(e) / r
                o["launches"] += 1; o["ms"] += ms; o["flops"] += f; o["bytes"] += b
                sh = o["shapes"].setdefault(d, [0, 0.0, 0.0, 0.0])
                sh[0] += 1; sh[1] += ms; sh[2] += f; sh[3] += b
        return out

    def summary(self):
        out = {}
        for kind, recs in self.records.items():
            ms = sum(r[0].elapsed_time(r[1]) / r[6] for r in recs)
            fl = sum(r[2] for r in recs)
            out[kind] = dict(launches=len(recs), ms=ms, flops=fl)
        return out


_timer = {"t": None}


def set_kernel_timer(t):
    _timer["t"] = t


def _timed(kind, flops, fn, desc=None, nbytes=0.0, kernel=None, idem=False):
    kt = _timer["t"]
    return fn() if kt is None else kt.run(kind, flops, fn, desc, nbytes, kernel() if callable(kernel) else kernel, idem)


def _tb(kind, nbytes, fn, desc=None, kernel=None):
    """HBM-bound launch: algorithmic bytes only (every operand moved once)."""
    kt = _timer["t"]
    return fn() if kt is None else kt.run(kind, 0.0, fn, desc, float(nbytes), kernel() if callable(kernel) else kernel)


def _bn_name(which, C, dt, act, flag):
    return lambda: L().rd_bn_kernel_name(which, C, dt, act, 1 if flag else 0).decode()


def _bn_finalize_apply(stats, y, bn, coef, z, pixels, C, act, slope, dt, st):
    """rd_bn_finalize_apply: coef rows = (scale, shift, mean, rstd) as rd_bn_finalize fills them"""
    lib = L()
    return _tb("bn_apply", 2 * y.numel() * y.element_size() + stats.numel() * 4,
               lambda: lib.rd_bn_finalize_apply(_p(stats), stats.shape[0], _p(y), _p(bn.weight.detach() if bn.weight is not None else None),
                                                _p(bn.bias.detach() if bn.bias is not None else None), float(bn.eps),
                                                float(bn.momentum if bn.momentum is not None else 0.1), _p(bn.running_mean), _p(bn.running_var),
                                                _p(coef[2]), _p(coef[3]), _p(coef[0]), _p(coef[1]), _p(z), pixels, C, act, slope, dt, st),
               "bn finalize+apply+act M=%d C=%d [one launch]" % (pixels, C), kernel=lambda: lib.rd_bn_slab_kernel_name(0, pixels, dt, act).decode())


def _bn_bwd_recompute(dz, z, y, mean, rstd, scale, shift, partial, coef2, dgam, dbet, acc, dy, dres, pixels, C, act, slope, dt, st, nbytes, desc):
    """rd_bn_act_bwd_recompute; under a kernel timer its reduce / finalize / apply launches are issued (and timed, and named) one by one so
    that bench.py's roofline can rank the BatchNorm passes next to the convolution kernels."""
    lib = L()
    if dres is None and lib.rd_bn_slab_ok(pixels, C, dt):      # wide layer on a small map: reduce + finalize + apply in one launch (rd_bn_slab.hip)
        return _tb("bn_backward", nbytes, lambda: lib.rd_bn_act_bwd_slab(_p(dz), _p(y), _p(mean), _p(rstd), _p(scale), _p(shift), _p(dgam), _p(dbet), acc, _p(dy),
                                                                       pixels, C, act, slope, dt, st),
                   desc + " [one launch]", kernel=lambda: lib.rd_bn_slab_kernel_name(1, pixels, dt, act).decode())
    args = (_p(dz), _p(z), _p(y), _p(mean), _p(rstd), _p(scale), _p(shift), _p(partial), _p(coef2), _p(dgam), _p(dbet), acc, _p(dy), _p(dres), pixels, C,
            act, slope, dt)
    if _timer["t"] is None:
        return lib.rd_bn_act_bwd_recompute(*args, st)
    rc = _tb("bn_backward", 2.0 * nbytes / 3.0, lambda: lib.rd_bn_act_bwd_recompute_phases(*args, 1, st), desc + " [reduce]", kernel=_bn_name(1, C, dt, act, True))
    rc = rc or _tb("bn_finalize", partial.numel() * 4, lambda: lib.rd_bn_act_bwd_recompute_phases(*args, 2, st), "bn bwd finalize C=%d" % C, kernel="bn_bwd_finalize_kernel")
    return rc or _tb("bn_backward", nbytes, lambda: lib.rd_bn_act_bwd_recompute_phases(*args, 4, st), desc + " [apply]", kernel=_bn_name(2, C, dt, act, True))


# ------------------------------------------------------------------------------------------------- tape
class Tape:
    def __init__(self):
        self.nodes = []
        self.grads = {}    # id(tensor) -> gradient tensor (same shape/dtype as the tensor)
        self.keep = {}     # id -> tensor, keeps ids unique while grads are pending
        self.req = set()   # ids of tensors that need a gradient
        self.pgrads = {}   # id(param) -> fp32 gradient tensor
        self.params = {}   # id(param) -> param
        self.grad_alloc = None  # optional callable(param) -> preallocated fp32 grad view (flat arena)
        self.deferred = []      # weight gradients of 1x1 / linear layers, issued as ONE grouped launch at the end of backward()
        self.conv_reduce = []   # (rd_wgrad_reduce_item, workspace) of convolution weight gradients whose split-K slabs are written:
        self.conv_reduce_w = set()   # ... summed by ONE rd_wgrad_reduce_batch launch per backward stage; ids of their weights
        self.colsum = []        # (rd_colsum_item, partial rows tensor, bias id): bias gradients finished by one launch per backward stage
        self.dw_reduce = []     # (rd_dw_wgrad_item, partial rows tensor, weight id): depthwise weight gradients, likewise
        self.ln_grads = {}      # id(gamma) -> dict(dg, db, acc, parts=[(partial rows tensor, rows)]): LayerNorm parameter gradients, likewise
        self.bn_src = {}        # id(output of conv -> BatchNorm -> act) -> dict(y, coef, act, slope, C, partial): lets the data gradient that
                                # produces its dz also produce the BatchNorm backward's sums (conv_block)

    def requires(self, *ts):
        return any(t is not None and id(t) in self.req for t in ts)

    def mark(self, t):
        self.req.add(id(t))
        self.keep[id(t)] = t

    def record(self, fn):
        self.nodes.append(fn)

    def add_grad(self, t, g):
        if t is None or g is None or id(t) not in self.req:
            return
        cur = self.grads.get(id(t))
        if cur is not None and isinstance(g, HeadGrad):      # a virtual gradient meets another contribution: written out the unfused way
            g = g.materialize(self)
        if isinstance(cur, HeadGrad):
            cur = cur.materialize(self)
        if cur is None:
            self.grads[id(t)] = g
        else:  # never in place: gradient tensors may be shared with other consumers
            s = torch.empty_like(cur)
            _chk(_tb("elementwise", 3 * cur.numel() * cur.element_size(),
                     lambda: L().rd_add(_p(cur), _p(g), _p(s), cur.numel(), rd_of(cur), _stream(cur)), "grad add"), "rd_add")
            self.grads[id(t)] = s

    def pop_grad(self, t):
        return self.grads.pop(id(t), None)

    def param_grad(self, p):
        """-> (fp32 grad tensor, accumulate flag).  First touch in this tape overwrites, later touches add."""
        g = self.pgrads.get(id(p))
        if g is not None:
            return g, 1
        g = self.grad_alloc(p) if self.grad_alloc is not None else None
        acc = 0
        if g is None:
            g = torch.empty_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
        elif p.grad is not None and p.grad.data_ptr() == g.data_ptr():
            acc = 1  # another region already wrote this step's gradient into the arena slot
        self.pgrads[id(p)] = g
        self.params[id(p)] = p
        return g, acc

    def backward(self):
        while self.backward_stage() is not None:
            pass

    def backward_stage(self):
        """Run backward nodes (newest first) up to and including the next stage mark; -> the mark's tag, or None when the tape is
        exhausted.  A driver that owns the calling thread (rcnet_main.GraphedStep) uses the boundaries to end one hipGraph capture and
        begin the next, and to hand finished gradient buckets to the all-reducer while the remaining backward is still queued."""
        while self.nodes:
            fn = self.nodes.pop()
            fn()
            tag = getattr(fn, "_stage", None)
            if tag is not None:
                return tag
        self.flush_deferred()
        self.keep = {}
        return None

    def flush_deferred(self):
        """rd_linear_wgrad_batch over every deferred (x, dy, weight): one launch + one ordered reduction instead of two small
        latency-bound launches per layer (RC-Net's LoFTR transformer alone has 96 of them per step)."""
        self.flush_conv_reduce()
        pending, self.deferred = self.deferred, []
        by_dt = {}
        for it in pending:                    # a tape may mix fp32 regions (RC-Net's point MLP) with bf16 ones: one batch per dtype
            by_dt.setdefault(it["x"].dtype, []).append(it)
        for items in by_dt.values():
            self._flush_items(items)

    def defer_conv_reduce(self, item, ws, weight):
        self.conv_reduce.append((item, ws))
        self.conv_reduce_w.add(id(weight))
        if len(self.conv_reduce) >= 48:
            self.flush_conv_reduce()

    def flush_conv_reduce(self):
        """One launch sums the split-K slabs of every convolution weight gradient produced since the last flush (rd_wgrad_reduce_batch)."""
        self.flush_colsum()
        self.flush_dw_reduce()
        self.flush_ln_grads()
        pending, self.conv_reduce, self.conv_reduce_w = self.conv_reduce, [], set()
        if not pending:
            return
        arr = (_lib.WgradReduceItem * len(pending))(*[it for it, _ in pending])
        ws0 = pending[0][1]
        nbytes = sum(ws.numel() * 4 for _, ws in pending)
        _chk(_tb("conv_wgrad", nbytes, lambda: L().rd_wgrad_reduce_batch(arr, len(pending), _stream(ws0)), "wgrad slab reduce batch n=%d" % len(pending)),
             "rd_wgrad_reduce_batch")

    def flush_dw_reduce(self):
        """One launch finishes every pending depthwise weight gradient (rd_dw_wgrad_finalize_batch)."""
        pending, self.dw_reduce = self.dw_reduce, []
        if not pending:
            return
        arr = (_lib.DwWgradItem * len(pending))(*[it for it, _, _ in pending])
        part0 = pending[0][1]
        _chk(_tb("conv_wgrad", sum(p_.numel() * 4 for _, p_, _ in pending),
                 lambda: L().rd_dw_wgrad_finalize_batch(arr, len(pending), _stream(part0)), "depthwise wgrad finalize batch n=%d" % len(pending)),
             "rd_dw_wgrad_finalize_batch")

    def defer_ln_grad(self, gamma, dg, db, acc, part, rows):
        e = self.ln_grads.get(id(gamma))
        if e is not None and len(e["parts"]) >= 4:      # an item holds four applications of one layer
            self.flush_ln_grads()
            e = None
        if e is None:
            e = self.ln_grads[id(gamma)] = dict(dg=dg, db=db, acc=acc, parts=[])
        e["parts"].append((part, rows))

    def flush_ln_grads(self):
        """One launch sums the LayerNorm parameter-gradient partials of every layer application since the last flush (rd_ln_grad_batch)."""
        pending, self.ln_grads = self.ln_grads, {}
        if not pending:
            return
        items = []
        for e in pending.values():
            it = _lib.LnGradItem()
            for i, (part, rows) in enumerate(e["parts"]):
                it.partial[i] = part.data_ptr(); it.rows[i] = rows
            it.dgamma, it.dbeta, it.C, it.nparts, it.accumulate = e["dg"].data_ptr(), e["db"].data_ptr(), e["dg"].numel(), len(e["parts"]), e["acc"]
            items.append(it)
        arr = (_lib.LnGradItem * len(items))(*items)
        any_part = next(iter(pending.values()))["parts"][0][0]
        _chk(L().rd_ln_grad_batch(arr, len(items), _stream(any_part)), "rd_ln_grad_batch")

    def flush_colsum(self):
        """One launch finishes every pending bias gradient (rd_colsum_finalize_batch)."""
        pending, self.colsum = self.colsum, []
        if not pending:
            return
        arr = (_lib.ColsumItem * len(pending))(*[it for it, _, _ in pending])
        part0 = pending[0][1]
        _chk(_tb("elementwise", sum(p_.numel() * 4 for _, p_, _ in pending),
                 lambda: L().rd_colsum_finalize_batch(arr, len(pending), _stream(part0)), "bias gradient finalize batch n=%d" % len(pending)),
             "rd_colsum_finalize_batch")

    def _flush_items(self, items):
        lib = L()
        groups = {}
        for it in items:                      # several uses of one weight share a reduction (fixed order: first use first)
            groups.setdefault(id(it["weight"]), []).append(it)
        total = 0
        for it in items:
            M = it["M"]
            if M > 40960:
                # enough (tile, split) blocks to fill the GPU, bounded by the slab traffic: every split writes (and the reduction
                # re-reads) a Cout x Cin slab -- at most 192 splits and 32 MB of slabs per item
                tci, tco = (it["Cin"] + 63) // 64, (it["Cout"] + 63) // 64      # blocks per split = groups of up to 3 + 1 / 2 x 2 tiles (lwg_shape, rd_linear_wgrad.hip)
                ni, no = (1, min(tco, 3)) if tci == 1 else ((min(tci, 3), 1) if tco == 1 else (2, 2))
                tiles = ((tci + ni - 1) // ni) * ((tco + no - 1) // no)
                # (round 6: 192 splits / ~384 blocks per item instead of 512 / ~1024 -- an item shares the launch with up to 32 others, and every split
                # writes a slab the reduction re-reads: SML 12.12 -> 12.06 ms in three alternating runs on one box, RC-Net unchanged)
                want = max(64, min(192, (384 + tiles - 1) // tiles, (32 << 20) // (4 * it["Cin"] * it["Cout"])))
                rps = max(128, ((M + want - 1) // want + 63) & ~63)
            else:
                rps = 640 if M >= 2560 else max(64, (M + 3) // 4 + 63 & ~63)
            it["rps"] = rps
            it["nsplit"] = (M + rps - 1) // rps
            it["off"] = total
            total += it["nsplit"] * it["Cout"] * it["Cin"]
        x0 = items[0]["x"]
        ws = torch.empty(total, dtype=torch.float32, device=x0.device)
        gem = (_lib.LwgGemm * len(items))()
        red = (_lib.LwgReduce * len(groups))()
        order, gi = [], 0
        for grp in groups.values():           # slabs of one group must be consecutive: lay the items out group by group
            order.extend(grp)
        off = 0
        for it in order:
            it["off"] = off
            off += it["nsplit"] * it["Cout"] * it["Cin"]
        for k, it in enumerate(order):
            g = gem[k]
            g.x1, g.x2, g.dy = it["x"].data_ptr(), (0 if it["x2"] is None else it["x2"].data_ptr()), it["dy"].data_ptr()
            g.slab = ws.data_ptr() + 4 * it["off"]
            g.M, g.C1, g.C2, g.Cout, g.nsplit, g.rows_per_split = it["M"], it["C1"], it["C2"], it["Cout"], it["nsplit"], it["rps"]
        for k, grp in enumerate(groups.values()):
            w = grp[0]["weight"]
            dw, acc = self.param_grad(w)
            r = red[k]
            r.slab, r.dw, r.elems = ws.data_ptr() + 4 * grp[0]["off"], dw.data_ptr(), grp[0]["Cout"] * grp[0]["Cin"]
            r.nsplit, r.accumulate = sum(i["nsplit"] for i in grp), acc
        flops = sum(i["flops"] for i in items)
        dt, st = rd_of(x0), _stream(x0)
        _chk(_timed("conv_wgrad", flops, lambda: lib.rd_linear_wgrad_batch(gem, len(items), red, len(groups), dt, st),
                    "wgrad grouped linear n=%d" % len(items)), "rd_linear_wgrad_batch")


def tape():
    return _state["tape"]


# ------------------------------------------------------------------------------------------ debug taps
_taps = {"on": False, "fwd": {}, "grad": {}}


def taps_enable(flag=True):
    """Debug / parity tooling (tools/grad_localise.py, tests): while on, `tap(name, t)` keeps a float copy of the tape tensor `t` and of the
    gradient the backward has accumulated for it when it reaches that point.  Off (the default) a tap is one dict lookup and records nothing."""
    _taps["on"] = bool(flag)
    _taps["fwd"], _taps["grad"] = {}, {}


def taps():
    return _taps["fwd"], _taps["grad"]


def tap(name, x):
    """Call right after `x` was produced (so that every consumer has contributed before the tap's backward node runs).  No-op unless taps_enable()."""
    if not _taps["on"]:
        return x
    t = tape()
    if not isinstance(x, LazyAct):      # (a virtual activation is not written for a tap: that would change the route under test)
        _taps["fwd"][name] = x.detach().float().cpu()
    if t is not None and id(x) in t.req:
        def backward():
            g = t.grads.get(id(x))
            if torch.is_tensor(g):
                _taps["grad"][name] = g.detach().float().cpu()
        t.record(backward)
    return x


# ------------------------------------------------------------------------------------------ stage marks
_stage_hooks = []


def add_stage_hook(fn):
    """fn(tag) is called from the backward whenever a stage mark is passed (gradients of everything recorded after the mark are final)."""
    _stage_hooks.append(fn)


def remove_stage_hook(fn):
    if fn in _stage_hooks:
        _stage_hooks.remove(fn)


def stage_mark(tag):
    """Record a backward-order boundary on the active tape.  When the backward reaches it, every operation recorded AFTER this call has
    run its backward, the grouped 1x1 / linear weight gradients collected so far are issued, and the stage hooks fire -- this is where
    data parallelism starts the all-reduce of the finished gradient bucket while earlier layers are still back-propagating
    (reference counterpart: torch.nn.DataParallel's gather of replica gradients, RCNet/rcnet_model.py:259-265)."""
    t = tape()
    if t is None:
        return

    def backward():
        t.flush_deferred()
        for fn in list(_stage_hooks):
            fn(tag)
    backward._stage = tag
    t.record(backward)


class StepTape(object):
    """One whole training step on ONE tape driven from the calling thread, without torch.autograd: module forwards called inside
    `forward()` run inline on this tape (engine.run_region), `seed()` sets dLoss, and `backward_stage()` walks the backward one stage
    at a time so that the caller can split hipGraph captures / start bucket all-reduces at the stage marks.  Parameter gradients land
    in the optimizer's arena views (or fresh fp32 tensors) and are deposited on `.grad` by `finish()`."""

    def __init__(self):
        self.tape = Tape()
        self.tape.grad_alloc = _grad_alloc_hook["fn"]

    def forward(self, fn):
        with torch.no_grad(), _active(self.tape):
            return fn()

    def seed(self, loss, value=1.0):
        t = self.tape
        if id(loss) not in t.req:
            raise RuntimeError("StepTape.seed: the loss does not depend on any trainable tensor of this tape")
        t.grads[id(loss)] = torch.full((1,), float(value), dtype=torch.float32, device=loss.device)

    def backward_stage(self):
        with torch.no_grad(), _active(self.tape):
            return self.tape.backward_stage()

    def finish(self):
        t = self.tape
        for pid, g in t.pgrads.items():
            t.params[pid].grad = g
        t.pgrads, t.params, t.grads = {}, {}, {}


@contextlib.contextmanager
def _active(t):
    prev = _state["tape"]
    _state["tape"] = t
    try:
        yield t
    finally:
        _state["tape"] = prev


_grad_alloc_hook = {"fn": None}
_grad_allocs = []      # weak references to the registered allocators (one per live flat-arena optimizer: RC-Net's and the SML's may coexist)


def _dispatch_grad_alloc(p):
    for ref in reversed(_grad_allocs):      # newest first; an allocator returns None for a parameter it does not own
        f = ref()
        if f is None:
            continue
        g = f(p)
        if g is not None:
            return g
    return None


def set_param_grad_allocator(fn):
    """fn(param) -> preallocated fp32 gradient tensor or None (used by the flat-arena optimizer / DDP).  Several allocators may be
    registered (two models with their own FlatAdam in one process): each is asked in turn, newest first; bound methods are held weakly,
    so an optimizer that is dropped takes its allocator with it.  None removes all of them."""
    if fn is None:
        del _grad_allocs[:]
        _grad_alloc_hook["fn"] = None
        return
    _grad_allocs[:] = [r for r in _grad_allocs if r() is not None]
    ref = weakref.WeakMethod(fn) if hasattr(fn, "__self__") else (lambda f=fn: f)
    _grad_allocs.append(ref)
    _grad_alloc_hook["fn"] = _dispatch_grad_alloc


class _Region(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, n_in, *tensors):
        inputs, params = tensors[:n_in], tensors[n_in:]
        t = Tape()
        t.grad_alloc = _grad_alloc_hook["fn"]
        for i, x in enumerate(inputs):
            if x is not None and ctx.needs_input_grad[2 + i]:
                t.mark(x)
        with _active(t):
            outs = runner(*inputs)
        single = not isinstance(outs, (tuple, list))
        outs = (outs,) if single else tuple(outs)
        ctx.tape, ctx.inputs, ctx.params, ctx.outs = t, inputs, params, outs
        ctx.set_materialize_grads(False)
        return outs[0] if single else outs

    @staticmethod
    def backward(ctx, *gouts):
        t = ctx.tape
        for o, g in zip(ctx.outs, gouts):
            if g is None or id(o) not in t.req:
                continue
            if g.dtype != o.dtype or g.stride() != o.stride():
                g = _match_layout(g, o)
            t.grads[id(o)] = g
        with _active(t):
            t.backward()
        gin = tuple(t.grads.get(id(x)) if x is not None else None for x in ctx.inputs)
        # Parameter gradients are deposited on .grad directly (returning them would make autograd's AccumulateGrad
        # clone every arena view: one copy kernel per parameter per step).
        for p in ctx.params:
            g = t.pgrads.get(id(p))
            if g is None:
                continue
            if p.grad is None or p.grad.data_ptr() == g.data_ptr():
                p.grad = g
            else:
                acc = torch.empty_like(g)
                _chk(L().rd_add(_p(p.grad), _p(g), _p(acc), g.numel(), RD_F32, _stream(g)), "rd_add")
                p.grad = acc
        ctx.tape = None
        return (None, None) + gin + (None,) * len(ctx.params)


def _match_layout(g, like):
    """Bring an incoming gradient to the dtype / strides of the tensor it belongs to (boundary only)."""
    out = torch.empty_strided(like.shape, like.stride(), dtype=like.dtype, device=like.device)
    if g.is_contiguous() and like.dim() == 4 and like.permute(0, 2, 3, 1).is_contiguous():
        n, c, h, w = like.shape
        _chk(L().rd_nchw_to_nhwc(_p(g), _p(out), n, c, h, w, rd_of(g), rd_of(out), 1.0, _stream(g)), "rd_nchw_to_nhwc")
    elif g.stride() == like.stride() or (g.is_contiguous() and like.is_contiguous()):
        _chk(L().rd_cast(_p(g), _p(out), g.numel(), rd_of(g), rd_of(out), 1.0, _stream(g)), "rd_cast")
    else:
        out.copy_(g)  # arbitrary user layout: torch copy, off the hot path
    return out


def run_region(runner, inputs, params, graph_key=None, on_replay=None):
    """Run `runner(*inputs)` as one autograd node, or inline when a tape is already active / grad is off.
    graph_key (hashable) names the launch sequence this call produces -- the module, its training flags, anything besides the input
    shapes that changes what is launched.  With engine.set_autograph(True) a region that is called again with the same key and input
    geometry is CAPTURED (forward and backward hipGraphs) and from then on replayed: this is how an unchanged training script -- eager
    torch.autograd, torch.optim.Adam -- gets the captured step's launch cost (see _graphed_region).  on_replay(): host-side bookkeeping a
    replay must repeat (BatchNorm num_batches_tracked counters)."""
    if _state["tape"] is not None:
        return runner(*inputs)
    if not torch.is_grad_enabled():
        with _active(None):
            return runner(*inputs)
    params = [p for p in params if p.requires_grad]
    if _autograph["on"] and graph_key is not None and _timer["t"] is None and not _stage_hooks and not _taps["on"]:
        out = _graphed_region(runner, inputs, params, graph_key, on_replay)
        if out is not _NOT_GRAPHED:
            return out
    return _Region.apply(runner, len(inputs), *inputs, *params)


# ------------------------------------------------------------------------------- captured regions (unchanged callers)
# The reference's training scripts drive the modules through torch.autograd: model.forward(...) -> compute_loss -> loss.backward() ->
# torch.optim.Adam.step() (RCNet/rcnet_main.py:342-359, train_zju.py:353-392).  Eagerly that is ~520 launches per RC-Net step issued from
# Python, and the host cannot issue them as fast as the GPU runs them (bench.py `unchanged_caller`: 434-712 img/s by box against
# 1090-1133 for the captured step).  rcnet_main.GraphedTrainStep needs the loop to change; this does not: a region whose key and input
# geometry repeat is captured ONCE -- its forward into one hipGraph, its whole backward (tape.backward() seeded with a static output
# gradient) into a second one sharing the pool -- and every later call copies the inputs into the static tensors, replays, and hands
# autograd the static outputs; the node's backward copies the output gradient in, replays, and deposits the static parameter gradients
# on .grad.  Same kernels, same order, same results as the eager region (tests compare them).  The first call with a key runs eagerly
# (one-off shapes never pay for a capture, packed operands and allocator are warm), the second captures.  As with
# torch.cuda.make_graphed_callables the outputs of a call are overwritten by the next call with the same key.
_autograph = {"on": False, "entries": {}, "max_entries": 4, "captured": 0, "replayed": 0, "eager": 0, "seen": {}}
_NOT_GRAPHED = object()


def set_autograph(flag, max_entries=4):
    """Capture and replay repeated autograd regions (default off).  INTEGRATION.md section 1: the aliasing block switches it on."""
    _autograph["on"] = bool(flag)
    _autograph["max_entries"] = int(max_entries)
    if not flag:
        clear_autograph()


def clear_autograph():
    _autograph["entries"].clear()
    _autograph["seen"].clear()


def autograph_stats():
    return {k: _autograph[k] for k in ("captured", "replayed", "eager")}


class _GraphEntry(object):
    __slots__ = ("g_f", "g_b", "static_in", "outs", "single", "static_gout", "gin", "pgrads", "params", "owner", "versions", "on_replay", "tape", "uses",
                 "flat_sg", "flat_pub", "pub", "offsets", "mine", "foreign")


def _geometry(x):
    return None if x is None else (tuple(x.shape), x.dtype, tuple(x.stride()), bool(x.requires_grad), x.device.index)


def _graphed_region(runner, inputs, params, graph_key, on_replay):
    if not all(x is None or (torch.is_tensor(x) and x.is_cuda) for x in inputs) or not params:
        return _NOT_GRAPHED
    key = (graph_key, _state["dtype"], tuple(_geometry(x) for x in inputs), tuple(id(p) for p in params))
    ag = _autograph
    e = ag["entries"].get(key)
    fresh = False
    if e is None:
        n = ag["seen"].get(key, 0)
        ag["seen"][key] = n + 1
        if n == 0:                    # first sight of this geometry: eager (and everything it allocates lazily is then in place)
            ag["eager"] += 1
            if len(ag["seen"]) > 64:
                ag["seen"].clear()
            return _NOT_GRAPHED
        e = _capture_region(runner, inputs, params, on_replay)
        while len(ag["entries"]) >= max(1, ag["max_entries"]):      # least recently used entry goes (its graphs and activations with it)
            victim = min(ag["entries"], key=lambda k: ag["entries"][k].uses)
            del ag["entries"][victim]
        ag["entries"][key] = e
        ag["captured"] += 1
        fresh = True
    ag["replayed"] += 1
    e.uses = ag["replayed"]
    return _GraphedRegion.apply(e, fresh, len(inputs), *inputs, *params)


def _capture_region(runner, inputs, params, on_replay):
    e = _GraphEntry()
    e.on_replay, e.params, e.uses = on_replay, {id(p): p for p in params}, 0
    e.owner = frozenset(id(p) for p in params)
    refresh_packed(e.owner)          # nothing is stale inside the capture: no pack launch gets recorded (and re-run by every replay)
    e.versions = sum(p._version for p in params)
    relaxed = torch.distributed.is_available() and torch.distributed.is_initialized()
    kw = {"capture_error_mode": "thread_local"} if relaxed else {}
    pool = torch.cuda.graph_pool_handle()
    hook = _grad_alloc_hook["fn"]
    # static parameter gradients: ONE flat fp32 buffer allocated before the capture (16-byte aligned slots, like the flat-arena optimizer's), and a
    # second one of the same layout that the caller sees on .grad -- publishing a backward is one device copy (or one add when the caller did
    # not clear the gradients), not one tensor operation per parameter
    offs, n = {}, 0
    for p_ in params:
        offs[id(p_)] = n
        n += (p_.numel() + 3) // 4 * 4
    e.flat_sg = torch.zeros(n, dtype=torch.float32, device=params[0].device)
    e.flat_pub = torch.zeros(n, dtype=torch.float32, device=params[0].device)
    e.pub = {id(p_): e.flat_pub[offs[id(p_)]:offs[id(p_)] + p_.numel()].view(p_.shape) for p_ in params}
    e.offsets = offs

    def alloc(p):      # a flat-arena optimizer's slot if there is one, else this entry's static slot
        g = hook(p) if hook is not None else None
        if g is None:
            o = offs[id(p)]
            g = e.flat_sg[o:o + p.numel()].view(p.shape)
        return g
    with torch.no_grad():
        e.static_in = [None if x is None else x.detach().clone() for x in inputs]
        t = Tape()
        t.grad_alloc = alloc
        for x, s_ in zip(inputs, e.static_in):
            if x is not None and x.requires_grad:
                t.mark(s_)
        torch.cuda.synchronize()
        e.g_f = torch.cuda.CUDAGraph()
        with torch.cuda.graph(e.g_f, pool=pool, **kw):
            with _active(t):
                outs = runner(*e.static_in)
        e.single = not isinstance(outs, (tuple, list))
        e.outs = (outs,) if e.single else tuple(outs)
        e.static_gout = [torch.zeros_like(o) if id(o) in t.req else None for o in e.outs]
        e.g_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(e.g_b, pool=pool, **kw):
            for o, g in zip(e.outs, e.static_gout):
                if g is not None:
                    t.grads[id(o)] = g
            with _active(t):
                t.backward()
            e.gin = [t.grads.get(id(s_)) if s_ is not None else None for s_ in e.static_in]
        e.pgrads = dict(t.pgrads)
        e.mine = [pid for pid, sg in e.pgrads.items() if sg.data_ptr() == e.flat_sg.data_ptr() + 4 * offs[pid]]      # written into this entry's flat buffer
        e.foreign = [pid for pid in e.pgrads if pid not in set(e.mine)]
        t.pgrads, t.grads = {}, {}
        e.tape = t        # (keeps the RoI overflow flag and anything else the tape owns alive with the graphs)
    return e


def _dense(t):
    """non-overlapping and dense: the strides are a permutation of a contiguous layout (contiguous, channels_last, permuted views of either)"""
    n = 1
    for size, stride in sorted(((sz, st) for sz, st in zip(t.shape, t.stride()) if sz != 1), key=lambda p: p[1]):
        if stride != n:
            return False
        n *= size
    return True


def dev_copy(dst, src):
    """dst <- src on the current stream with a KERNEL (rd_cast) where the two tensors have the same dense layout and a dtype the library knows;
    torch's copy_ otherwise.  Used for everything that is copied right in front of a hipGraph replay: torch's contiguous same-dtype copy is a
    hipMemcpyAsync, and on this ROCm a memory node / copy next to graph kernels has been seen mis-ordered once already (the
    hipMemsetAsync finding; round 6: gradients cloned right before a backward replay came out corrupted) -- kernel -> graph ordering is the
    path every captured step has exercised since round 1."""
    if (dst.dtype == src.dtype and dst.dtype in _RD_DT and dst.shape == src.shape and dst.stride() == src.stride() and src.is_cuda and dst.is_cuda
            and dst.numel() > 0 and _dense(dst)):
        _chk(L().rd_cast(_p(src), _p(dst), src.numel(), rd_of(src), rd_of(dst), 1.0, _stream(src)), "rd_cast")
    else:
        dst.copy_(src, non_blocking=True)


class _GraphedRegion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, fresh, n_in, *tensors):
        for s_, x in zip(e.static_in, tensors[:n_in]):
            if s_ is not None and x.data_ptr() != s_.data_ptr():
                dev_copy(s_, x)
        if not fresh:
            v = sum(p._version for p in e.params.values())
            if v != e.versions:       # an optimizer rewrote the parameters through torch since the last replay: ONE launch re-packs this region's operands
                refresh_packed(e.owner)
                e.versions = v
            if e.on_replay is not None:
                e.on_replay()
        e.g_f.replay()
        ctx.e = e
        ctx.set_materialize_grads(False)
        outs = tuple(o.detach() for o in e.outs)
        return outs[0] if e.single else outs

    @staticmethod
    def backward(ctx, *gouts):
        e = ctx.e
        for sg, g in zip(e.static_gout, gouts):
            if sg is None:
                continue
            if g is None:
                sg.zero_()
            else:
                dev_copy(sg, g)
        e.g_b.replay()
        # publish: the caller's .grad tensors are views of e.flat_pub.  Cleared gradients (zero_grad(set_to_none), the reference loop's habit) ->
        # one copy of the whole static buffer; gradients still in place from the previous backward -> one add (torch's accumulation semantics);
        # anything else the caller put on .grad is added to parameter by parameter.
        mine = e.mine
        state = [e.params[pid].grad for pid in mine]
        if all(g is None for g in state):
            dev_copy(e.flat_pub, e.flat_sg)
            for pid in mine:
                e.params[pid].grad = e.pub[pid]
        elif all(g is not None and g.data_ptr() == e.pub[pid].data_ptr() for g, pid in zip(state, mine)):
            e.flat_pub.add_(e.flat_sg)
        else:
            for g, pid in zip(state, mine):
                p = e.params[pid]
                p.grad = e.pgrads[pid].clone() if g is None else g + e.pgrads[pid]
        for pid in e.foreign:      # slots of a flat-arena optimizer: written in place by the replay, as in the eager region
            p, sg = e.params[pid], e.pgrads[pid]
            if p.grad is None or p.grad.data_ptr() == sg.data_ptr():
                p.grad = sg
            else:
                p.grad = p.grad + sg
        n_in = len(e.static_in)
        return (None, None, None) + tuple(e.gin) + (None,) * (len(ctx.needs_input_grad) - 3 - n_in)


# --------------------------------------------------------------------------------------- small helpers
def empty(shape, like, dtype=None):
    return torch.empty(shape, dtype=dtype or like.dtype, device=like.device)


def to_act(x, scale=1.0):
    """Any fp32/bf16 contiguous tensor -> engine activation dtype (same shape), optionally scaled."""
    dt = act_dtype()
    if x.dtype == dt and scale == 1.0:
        return x
    out = torch.empty(x.shape, dtype=dt, device=x.device)
    _chk(L().rd_cast(_p(x), _p(out), x.numel(), rd_of(x), rd_of(out), float(scale), _stream(x)), "rd_cast")
    return out


def cast(x, dtype, scale=1.0):
    if x.dtype == dtype and scale == 1.0:
        return x
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _chk(L().rd_cast(_p(x), _p(out), x.numel(), rd_of(x), rd_of(out), float(scale), _stream(x)), "rd_cast")
    return out


def nchw_to_nhwc(x, scale=1.0, dtype=None):
    """Logical NCHW tensor (any of: contiguous, channels_last) -> engine (N,H,W,C) activation tensor."""
    n, c, h, w = x.shape
    dt = dtype or act_dtype()
    xp = x.permute(0, 2, 3, 1)
    if xp.is_contiguous():
        return cast(xp, dt, scale) if (x.dtype != dt or scale != 1.0) else xp
    if not x.is_contiguous():
        x = x.contiguous()
    out = torch.empty((n, h, w, c), dtype=dt, device=x.device)
    _chk(L().rd_nchw_to_nhwc(_p(x), _p(out), n, c, h, w, rd_of(x), rd_of(out), float(scale), _stream(x)), "rd_nchw_to_nhwc")
    return out


def as_nchw(x):
    """(N,H,W,C) engine tensor -> logical NCHW view (channels_last strides, zero copy)."""
    return x.permute(0, 3, 1, 2)


# packed-weight cache: (id(param), version, mode, dtype) -> tensor
_pack_cache = {}


def packed_weight(w, mode, dt, cin_pad=0):
    """MFMA-layout copy of a weight tensor, re-packed whenever the tensor was written (version counter) or re-allocated.
    Entries die with their tensor (weakref callback): CPython recycles id()s and the caching allocator recycles addresses, so an
    (id, data_ptr, version) key alone can match a DIFFERENT later tensor -- seen as stale weights / out-of-bounds reads in stress runs.
    A stale entry is re-packed INTO ITS EXISTING BUFFER: captured hipGraphs keep reading that address, so a weight written through
    torch (load_state_dict, broadcast) must never move its packed operand."""
    key = (w._version, mode, dt, w.data_ptr(), tuple(w.shape))
    slot = (mode, dt) if not cin_pad else (mode, dt, cin_pad)
    ent = _pack_cache.get(id(w))
    if ent is not None and ent["ref"]() is not w:
        ent = None
    if ent is None:
        wid = id(w)
        ent = {"ref": weakref.ref(w, lambda _r, wid=wid: _pack_cache.pop(wid, None))}
        _pack_cache[wid] = ent
    hit = ent.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    cout, cin, kh, kw = w.shape if w.dim() == 4 else (w.shape[0], w.shape[1], 1, 1)
    if cin_pad:     # forward operand with zero-padded input channels (3-channel stems on the vector kernels)
        assert mode == 0 and cin_pad >= cin
        n = L().rd_conv_packed_elems(cout, kh * kw * cin_pad, dt)
    else:
        rows, c = (4 * cout, cin) if mode == 2 else ((cin, 4 * cout) if mode == 3 else ((cin, cout) if mode else (cout, cin)))
        n = L().rd_conv_packed_elems(rows, kh * kw * c, dt)
    buf = hit[1] if (hit is not None and hit[1].numel() == n and hit[1].device == w.device) else torch.empty(n, dtype=_TORCH_DT[dt], device=w.device)
    if cin_pad:
        _chk(L().rd_conv_pack_weights_padded(_p(w.detach()), _p(buf), cout, cin, cin_pad, kh, kw, dt, _stream(w)), "rd_conv_pack_weights_padded")
    else:
        _chk(L().rd_conv_pack_weights(_p(w.detach()), _p(buf), cout, cin, kh, kw, mode, dt, _stream(w)), "rd_conv_pack_weights")
    ent[slot] = (key, buf)
    return buf


def clear_caches():
    _pack_cache.clear()
    _pack_table.clear()


_pack_table = {}      # owner key -> dict(sig, dev, n, keep): the device-side item table of one rd_conv_pack_weights_batch launch


def refresh_packed(owner=None):
    """Re-pack cached operands in one launch (rd_conv_pack_weights_batch).  Called by the optimizer right after it rewrote the
    parameters in place: the cached buffers keep their addresses (hipGraph replays keep reading them) and the next forward finds
    every operand fresh instead of issuing one pack launch per layer and direction.
    owner: a frozenset of id(parameter) -- only those parameters' operands (a FlatAdam passes its own: with two models in one process, RC-Net's
    step does not re-pack the Scale Map Learner's weights); None: every cached operand."""
    live = []
    for wid, ent in list(_pack_cache.items()):
        if owner is not None and wid not in owner:
            continue
        w = ent["ref"]()
        if w is None:
            continue
        for k, hit in list(ent.items()):
            if k == "ref":
                continue
            mode, dt = k[0], k[1]
            cpad = k[2] if len(k) > 2 else 0
            key, buf = hit
            cur = (w._version, mode, dt, w.data_ptr(), tuple(w.shape))
            if key != cur:
                if key[4] != cur[4] or buf.device != w.device:
                    ent.pop(k)      # re-shaped / moved to another device: the next forward packs a fresh operand
                    continue
                ent[k] = (cur, buf)  # written through torch (load_state_dict, broadcast) since it was packed: re-packed below IN PLACE
            live.append((w, buf, mode, dt, cpad))
    if not live:
        return
    sig = tuple((w.data_ptr(), buf.data_ptr(), mode, dt, cpad) for w, buf, mode, dt, cpad in live) + (_state["pack_vec"], _state["pack_map"])
    halves = {dt for _, _, _, dt, _ in live if dt != RD_F32}
    if len(halves) > 1:
        raise RuntimeError("cached packed weights mix bf16 and fp16 operands: call engine.clear_caches() when switching the compute dtype")
    half = halves.pop() if halves else RD_BF16
    tab = _pack_table.setdefault(owner, {})
    if tab.get("sig") != sig:
        items = (_lib.PackItem * len(live))()
        for it, (w, buf, mode, dt, cpad) in zip(items, live):
            cout, cin, kh, kw = w.shape if w.dim() == 4 else (w.shape[0], w.shape[1], 1, 1)
            # inside the table the item dtype is 0 (fp32) or 1 (the 16-bit type named by `half`, rd_conv_pack_weights_batch_half)
            it.w, it.packed, it.Cout, it.Cin, it.KH, it.KW, it.mode, it.dtype = w.data_ptr(), buf.data_ptr(), cout, (cpad or cin), kh, kw, mode, (0 if dt == RD_F32 else 1)
            it.Cin_src = cin if cpad else 0
            it.reserved = 0 if _state["pack_vec"] else 1
        host = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        tab["dev"] = host.to(live[0][0].device)
        tab["sig"] = sig
        tab["n"] = len(live)
        tab["keep"] = [b for _, b, _, _, _ in live]
        # block map: block -> (item, block of the item, blocks of the item): the launch has the blocks the operands need, not 256 per item
        bmap = []
        for i, (w, buf, mode, dt, cpad) in enumerate(live):
            nb = _pack_blocks(w, mode, dt, cpad)
            bmap += [(i, b, nb, 0) for b in range(nb)]
        tab["map"] = torch.tensor(bmap, dtype=torch.int32).to(live[0][0].device)
        tab["blocks"] = len(bmap)
    w0 = live[0][0]
    if _state["pack_map"]:
        _chk(L().rd_conv_pack_weights_batch_map(_p(tab["dev"]), tab["n"], half, _p(tab["map"]), tab["blocks"], _stream(w0)), "rd_conv_pack_weights_batch_map")
    else:
        _chk(L().rd_conv_pack_weights_batch_half(_p(tab["dev"]), tab["n"], half, _stream(w0)), "rd_conv_pack_weights_batch_half")


def _pack_blocks(w, mode, dt, cpad):
    """256-thread blocks one packed operand gets in the batched re-pack: one thread per work item of the 16-byte-unit form (a tile of 8 rows x 8
    channel groups per wave; csrc/rd_conv.hip pack_one_vec) or per element of the element-wise form; 1 .. 256.  Sizing only: the kernel's loops
    cover an operand with any number of blocks."""
    cout, cin, kh, kw = w.shape if w.dim() == 4 else (w.shape[0], w.shape[1], 1, 1)
    if cpad:
        cin = cpad
    ve = 4 if dt == RD_F32 else 8
    rows = 4 * cout if mode == 2 else (cin if mode else cout)
    C = 4 * cout if mode == 3 else (cout if mode == 1 else cin)
    kk = kh * kw
    vec = _state["pack_vec"] and (kk == 1 or (kh == 3 and kw == 3)) and not cpad and C % ve == 0 and cin % 4 == 0 and (mode not in (2, 3) or (kk == 9 and cout % ve == 0))
    if vec:
        work = ((rows + 7) // 8) * ((C // ve + 7) // 8) * 64
    else:
        bn = 16 if rows <= 16 else (32 if rows <= 32 else (64 if rows <= 64 else 128))
        bke = 32 if dt == RD_F32 else 64
        work = ((rows + bn - 1) // bn * bn) * ((kk * C + bke - 1) // bke * bke)
        work = (work + 3) // 4      # the element-wise form strides: four elements per thread
    return max(1, min(256, (work + 255) // 256))


# ------------------------------------------------------------------------------------------ virtual activations
class LazyAct(object):
    """z = act(scale[c] * y + shift[c]) of a BatchNorm-ed convolution that is NOT written to HBM: the raw convolution output `y`, the
    per-channel coefficients of rd_bn_finalize and the activation code (reference: utils/net_utils.py:84-91 conv -> BatchNorm2d -> act).
    conv_block (forward staging and weight-gradient staging of the 3x3 kernels) and add_act apply the map while they read y -- rounded
    exactly as rd_affine_act would have stored z, so every result is bit-identical to the materialised path -- and `materialize()` writes
    z (once) for any consumer or kernel route that cannot.  The object itself is the tape key of z: gradients are added to it and the
    producing conv_block pops them."""
    __slots__ = ("y", "coef", "act", "slope", "_z", "__weakref__")

    def __init__(self, y, coef, act, slope):
        self.y, self.coef, self.act, self.slope, self._z = y, coef, act, float(slope), None

    shape = property(lambda self: self.y.shape)
    dtype = property(lambda self: self.y.dtype)
    device = property(lambda self: self.y.device)

    def fusion(self):
        f = _lib.ConvFusion()
        f.in_scale, f.in_shift, f.in_act, f.in_slope = self.coef[0].data_ptr(), self.coef[1].data_ptr(), self.act, self.slope
        return f

    def materialize(self):
        if self._z is None:
            lazy_counts["materialized"] += 1
            y = self.y
            C = y.shape[-1]
            z = torch.empty_like(y)
            nb = y.numel() * y.element_size()
            _chk(_tb("bn_apply", 2 * nb, lambda: L().rd_affine_act(_p(y), _p(self.coef[0]), _p(self.coef[1]), None, _p(z), y.numel() // C, C, self.act,
                                                                    self.slope, rd_of(y), _stream(y)), "bn apply+act M=%d C=%d" % (y.numel() // C, C),
                     kernel=_bn_name(0, C, rd_of(y), self.act, False)), "rd_affine_act")
            self._z = z
        return self._z


class HeadGrad(object):
    """The gradient of a LazyAct whose consumer is the one-channel 3x3 output convolution (conv_block's head route): dlogits + the head's
    weight stand in for the 16-channel tensor, which the producer's backward never needs in HBM (rd_bn_head_bwd_reduce / _apply recompute it
    from the nine neighbouring dlogits of a pixel).  materialize(): the unfused backward of the head convolution, for a producer that
    cannot take the virtual form."""
    __slots__ = ("lz", "dl", "weight", "w_req", "geom", "flops", "shp")

    def __init__(self, lz, dl, weight, w_req, geom, flops, shp):
        self.lz, self.dl, self.weight, self.w_req, self.geom, self.flops, self.shp = lz, dl, weight, w_req, geom, flops, shp

    def materialize(self, t):
        lib = L()
        lazy_counts["head_unfused_bwd"] += 1
        N, H, W, C, dt = self.geom
        dl, st = self.dl, _stream(self.dl)
        if self.w_req:
            a = self.lz.materialize()
            d = _desc(dt, N, H, W, C, 0, False, H, W, 1, 3, 3, 1, 1, 1, H, W, ACT_NONE, 0.0, 1)
            dw, acc = t.param_grad(self.weight)
            ws = torch.empty(lib.rd_conv_wgrad_workspace_bytes(ctypes.byref(d)) // 4, dtype=torch.float32, device=dl.device)
            _chk(_timed("conv_wgrad", self.flops, lambda: lib.rd_conv_wgrad(ctypes.byref(d), _p(a), None, _p(dl), _p(ws), _p(dw), acc, st), "wgrad " + self.shp),
                 "rd_conv_wgrad")
        dd = _desc(dt, N, H, W, 1, 0, False, H, W, C, 3, 3, 1, 1, 1, H, W, ACT_NONE, 0.0, C)
        da = torch.empty((N, H, W, C), dtype=dl.dtype, device=dl.device)
        _chk(_timed("conv_gemm", self.flops, lambda: lib.rd_conv_fwd(ctypes.byref(dd), _p(dl), None, _p(packed_weight(self.weight, 1, dt)), None, _p(da), None,
                                                                        None, st), "dgrad " + self.shp), "rd_conv_fwd(dgrad)")
        return da


def _head_name(which, dt, act):
    return lambda: L().rd_bn_head_kernel_name(which, dt, act).decode()


def materialize(x):
    """LazyAct -> its activated tensor (written on first use); tensors pass through."""
    return x.materialize() if isinstance(x, LazyAct) else x


# ------------------------------------------------------------------------------------------ conv block
def _desc(dt, N, Hin, Win, C1, C2, up, H1, W1, Cout, KH, KW, stride, pad, dil, OH, OW, act, slope, D1):
    d = ConvDesc()
    d.dtype, d.N, d.Hin, d.Win, d.C1, d.C2 = dt, N, Hin, Win, C1, C2
    d.upsample, d.H1, d.W1 = (1 if up else 0), H1, W1
    d.Cout, d.KH, d.KW, d.stride, d.pad, d.in_dilate = Cout, KH, KW, stride, pad, dil
    d.OH, d.OW, d.act, d.slope, d.D1 = OH, OW, act, slope, D1
    return d


def _conv_head(lz, xk, weight, N, H, W, C, dt, st):
    """conv_block's head route: logits = conv3x3(act(BN(y))) with one output channel, straight from the producer's raw output y
    (rd_bn_head_fwd: the activated tensor is never written); the backward hands the producer a HeadGrad instead of a tensor."""
    lib = L()
    t = tape()
    y = lz.y
    es = y.element_size()
    logits = torch.empty((N, H, W, 1), dtype=y.dtype, device=y.device)
    flops = 2.0 * N * H * W * 9 * C
    shp = "M=%d Cin=%d Cout=1 k=3 s=1" % (N * H * W, C)
    lazy_counts["head_fused"] += 1
    w32 = weight.detach()
    _chk(_timed("conv_gemm", flops, lambda: lib.rd_bn_head_fwd(_p(y), _p(lz.coef[0]), _p(lz.coef[1]), lz.act, lz.slope, _p(w32), _p(logits), N, H, W, C, dt, st),
                "fwd " + shp + " (bn+head)", y.numel() * es + N * H * W * es, kernel=_head_name(0, dt, lz.act), idem=True), "rd_bn_head_fwd")
    if t is None:
        return logits
    w_req = weight.requires_grad
    t.mark(logits)

    def backward():
        dl = t.pop_grad(logits)
        if dl is None:
            return
        if not dl.is_contiguous():
            dl = dl.contiguous()
        t.add_grad(xk, HeadGrad(lz, dl, weight, w_req, (N, H, W, C, dt), flops, shp))

    t.record(backward)
    return logits


def conv_block(x, weight, *, x2=None, bias=None, stride=1, pad=None, up=None, bn=None, act=ACT_NONE, slope=0.2,
               residual=None, training=True, out_hw=None, lazy_out=False):
    """act(BN(conv([up(x) | up(x2)], weight) + bias) + residual) on NHWC tensors; records its own backward.

    x: (N,H,W,C1) [, x2: (N,H,W,C2)]; weight OIHW fp32 (nn.Linear [out,in] is treated as 1x1);
    up: (Hv, Wv) nearest-upsample target applied to the sources inside the gather;
    bn: a torch.nn.BatchNorm2d used as a parameter container (train: batch stats + running update).
    x may be a LazyAct (the un-materialised output of a BatchNorm-ed conv_block): the kernels apply its scale / shift / activation while
    staging; lazy_out = 1 / 2 returns this layer's output as a LazyAct (callers whose consumers are conv_block / add_act) when the engine's
    lazy_bn level (set_lazy_bn) is at least that.
    """
    lib = L()
    t = tape()
    xk = x                      # tape key of the first source (the LazyAct itself when the input is virtual)
    lz = x if isinstance(x, LazyAct) else None
    if lz is not None:
        x = lz.y
    dt = rd_of(x)
    st = _stream(x)
    N, H1, W1, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[3]
    if weight.dim() == 4:
        Cout, Cin, KH, KW = weight.shape
    else:
        (Cout, Cin), KH, KW = weight.shape, 1, 1
    assert Cin == C1 + C2, "conv_block: weight expects %d input channels, got %d" % (Cin, C1 + C2)
    if pad is None:
        pad = KH // 2
    Hin, Win = (int(up[0]), int(up[1])) if up is not None else (H1, W1)
    is_up = up is not None and (Hin, Win) != (H1, W1)
    OH = (Hin + 2 * pad - KH) // stride + 1
    OW = (Win + 2 * pad - KW) // stride + 1
    if out_hw is not None:  # asymmetric (TF-"SAME") padding: `pad` is the leading pad, the output size is given
        OH, OW = int(out_hw[0]), int(out_hw[1])
    use_bn = bn is not None
    conv_act = ACT_NONE if (use_bn or residual is not None) else act
    # few-channel inputs (the 3-channel image stems) are zero-padded to one 16-byte vector so that forward and weight gradient run on
    # the vector / MFMA-bf16 kernels instead of the scalar-gather fallbacks (7x7 stem: 0.31 + 0.52 ms per RC-Net step)
    ve = 16 // x.element_size()
    cin_pad = 0
    if x2 is None and not is_up and C1 % ve != 0 and KH * KW >= 9 and not lib.rd_conv_fwd_streams(
            ctypes.byref(_desc(dt, N, Hin, Win, C1, 0, False, H1, W1, Cout, KH, KW, stride, pad, 1, OH, OW, conv_act, slope, Cout))):
        # (layers the streaming few-channel kernels take -- SML's 3 -> 3 `first` convolution -- are handed over as they are; an input that
        # needs a gradient -- the SML backbone's stem behind `first` -- gets it from the un-padded descriptor below)
        cin_pad = (C1 + ve - 1) // ve * ve
        xp = torch.empty((N, H1, W1, cin_pad), dtype=x.dtype, device=x.device)
        _chk(lib.rd_pad_channels(_p(x), _p(xp), N * H1 * W1, C1, cin_pad, dt, st), "rd_pad_channels")
        x_real, x, C1_real, C1 = x, xp, C1, cin_pad
    if (lz is not None and _state["bn_head"] and lz._z is None and KH == 3 and KW == 3 and stride == 1 and pad == 1 and Cout == 1 and x2 is None
            and not is_up and bias is None and not use_bn and residual is None and act == ACT_NONE and out_hw is None and not cin_pad
            and weight.dtype == torch.float32 and weight.is_contiguous() and (t is None or t.requires(xk))
            and lib.rd_bn_head_ok(N, H1, W1, C1, dt)):
        return _conv_head(lz, xk, weight, N, H1, W1, C1, dt, st)
    d = _desc(dt, N, Hin, Win, C1, C2, is_up, H1, W1, Cout, KH, KW, stride, pad, 1, OH, OW, conv_act, slope, Cout)
    fus = None
    if lz is not None:          # virtual input: fused where the kernel this shape is routed to stages whole channel vectors, else z is written now
        fus = lz.fusion()
        if cin_pad or not lib.rd_conv_fusion_ok(ctypes.byref(d), ctypes.byref(fus)):
            assert not cin_pad
            fus, x = None, lz.materialize()
    # exact-2x nearest up-sampling + 3x3 (UpConv2d at 15x6 -> 30x12 ... 120x50 -> 240x100): per output parity class (a, b) it is a 2x2 convolution of
    # the SOURCE with pre-summed taps.  Where the library has that form (rd_conv_up2_ok) the forward runs at source resolution with 4 Cout output
    # channels = (class, channel), skips the structurally zero (tap, class) blocks and stores depth-to-space: 2.25 x fewer MACs, a quarter of the
    # staged pixels.  The backward keeps the virtual-resolution form (same function; the pre-summed weights are rounded once more in bf16).
    d_up2 = None
    if (is_up and C2 == 0 and (Hin, Win) == (2 * H1, 2 * W1) and KH == 3 and KW == 3 and stride == 1 and pad == 1 and conv_act == ACT_NONE and bias is None and lz is None
            and not cin_pad and out_hw is None and residual is None and _state["up2_on_source"]):
        d_up2 = _desc(dt, N, H1, W1, C1, 0, False, H1, W1, 4 * Cout, 3, 3, 1, 1, 1, H1, W1, ACT_NONE, slope, Cout)
        d_up2.out_d2s = 1
        if not lib.rd_conv_up2_ok(ctypes.byref(d_up2)):
            d_up2 = None
    wp = packed_weight(weight, 2 if d_up2 is not None else 0, dt, cin_pad)
    y = torch.empty((N, OH, OW, Cout), dtype=x.dtype, device=x.device)
    stats = None
    bn_train = use_bn and (training or not bn.track_running_stats)
    if bn_train:
        if d_up2 is not None:      # rows of 4 Cout columns = four rows of Cout
            stats = torch.empty((4 * lib.rd_conv_stats_rows(ctypes.byref(d_up2)), Cout, 2), dtype=torch.float32, device=x.device)
        else:
            rows = lib.rd_conv_stats_rows(ctypes.byref(d))
            stats = torch.empty((rows, Cout, 2), dtype=torch.float32, device=x.device)
    flops = 2.0 * N * OH * OW * Cout * KH * KW * Cin  # algorithmic (2 FLOP per MAC), same count for dgrad / wgrad
    shp = "M=%d Cin=%d Cout=%d k=%d s=%d%s" % (N * OH * OW, Cin, Cout, KH, stride, " up" if is_up else "")
    bias_t = bias.detach() if bias is not None else None
    es = x.element_size()   # algorithmic HBM bytes of the three convolution launches (every operand moved exactly once)
    b_in = (x.numel() + (0 if x2 is None else x2.numel())) * es
    b_w, b_out = weight.numel() * es, N * OH * OW * Cout * es
    res_fused = False
    if (residual is not None and not use_bn and act == ACT_NONE and fus is None and residual.shape == y.shape and residual.dtype == y.dtype
            and residual.is_contiguous() and _state.get("fuse_res_add", True) and lib.rd_conv_add_ok(ctypes.byref(d))):
        # conv + bias + residual with no BatchNorm and no activation (the residual units of the SML decoder, modules/midas/blocks.py:99-130):
        # the residual rides in the convolution's epilogue as its addend (rounded once) instead of a separate add pass over y
        res_fused = True
        _chk(_timed("conv_gemm", flops, lambda: lib.rd_conv_fwd_add(ctypes.byref(d), _p(x), _p(x2), _p(wp), _p(bias_t), _p(residual), _p(y), st),
                    "fwd " + shp + " (+res)", b_in + b_w + 2 * b_out,
                    kernel=lambda: lib.rd_conv_fwd_kernel_name(ctypes.byref(d)).decode(), idem=True), "rd_conv_fwd_add")
    elif d_up2 is not None:
        lazy_counts["up2_fwd"] += 1
        _chk(_timed("conv_gemm", flops * 4.0 / 9.0, lambda: lib.rd_conv_fwd(ctypes.byref(d_up2), _p(x), None, _p(wp), None, _p(y), None, _p(stats), st),
                    "fwd " + shp + " (on source)", b_in + b_w + b_out,
                    kernel=lambda: lib.rd_conv_fwd_kernel_name(ctypes.byref(d_up2)).decode(), idem=True), "rd_conv_fwd(up2)")
    elif fus is not None:
        lazy_counts["fwd_fused"] += 1
        _chk(_timed("conv_gemm", flops, lambda: lib.rd_conv_fwd_fused(ctypes.byref(d), ctypes.byref(fus), _p(x), _p(x2), _p(wp), _p(bias_t), None,
                                                                         _p(y), None, _p(stats), st), "fwd " + shp + " (bn-in)", b_in + b_w + b_out,
                    kernel=lambda: lib.rd_conv_fused_kernel_name(ctypes.byref(d), ctypes.byref(fus)).decode(), idem=True), "rd_conv_fwd_fused")
    else:
        _chk(_timed("conv_gemm", flops, lambda: lib.rd_conv_fwd(ctypes.byref(d), _p(x), _p(x2), _p(wp), _p(bias_t), _p(y), None,
                                                                   _p(stats), st), "fwd " + shp, b_in + b_w + b_out,
                    kernel=lambda: lib.rd_conv_fwd_kernel_name(ctypes.byref(d)).decode(), idem=True), "rd_conv_fwd")
    pixels = N * OH * OW
    scale = shift = mean = rstd = None
    slab = bool(use_bn and bn_train and stats is not None and residual is None and not res_fused and lib.rd_bn_slab_ok(pixels, Cout, dt)
                and not (lazy_out and _state["lazy_bn"] >= int(lazy_out) and _state["bn_recompute"] and Cout % ve == 0))
    if use_bn:
        coef = torch.empty((4, Cout), dtype=torch.float32, device=x.device)
        scale, shift, mean, rstd = coef[0], coef[1], coef[2], coef[3]
    if slab:      # wide layer on a small map: finalize + apply in one launch (rd_bn_slab.hip)
        z = torch.empty_like(y)
        _chk(_bn_finalize_apply(stats, y, bn, coef, z, pixels, Cout, act, slope, dt, st), "rd_bn_finalize_apply")
    elif use_bn:
        _chk(_tb("bn_finalize", 0 if stats is None else stats.numel() * 4,
                 lambda: lib.rd_bn_finalize(_p(stats), 0 if stats is None else stats.shape[0], Cout, float(pixels),
                                            _p(bn.weight.detach() if bn.weight is not None else None),
                                            _p(bn.bias.detach() if bn.bias is not None else None), float(bn.eps),
                                            float(bn.momentum if bn.momentum is not None else 0.1), 1 if bn_train else 0,
                                            _p(bn.running_mean), _p(bn.running_var), _p(mean), _p(rstd), _p(scale), _p(shift), st),
                 "bn finalize C=%d" % Cout), "rd_bn_finalize")
    lazy = None
    if lazy_out and use_bn and residual is None and _state["lazy_bn"] >= int(lazy_out) and _state["bn_recompute"] and Cout % ve == 0:
        z = None
        lazy = LazyAct(y, coef, act, slope)      # z stays virtual: the consumer applies (scale, shift, act) while it stages y
    elif res_fused:
        z = y
    elif slab:
        pass
    elif use_bn or residual is not None:
        z = torch.empty_like(y)
        _chk(_tb("bn_apply", (2 + (residual is not None)) * b_out,
                 lambda: lib.rd_affine_act(_p(y), _p(scale), _p(shift), _p(residual), _p(z), pixels, Cout, act, slope, dt, st),
                 "bn apply+act M=%d C=%d" % (pixels, Cout), kernel=_bn_name(0, Cout, dt, act, residual is not None)), "rd_affine_act")
    else:
        z = y
    zk = lazy if lazy is not None else z      # what the caller receives = the tape key of this layer's output
    if t is None:
        return zk

    need_in = t.requires(xk, x2)
    need_res = t.requires(residual)
    w_req = weight.requires_grad
    if not (need_in or need_res or w_req or (use_bn and bn.weight is not None and bn.weight.requires_grad)):
        return zk
    t.mark(zk)
    if use_bn and bn_train and residual is None and _state["bn_recompute"] and _state["bn_bwd_fused"] and Cout % ve == 0:
        t.bn_src[id(zk)] = dict(y=y, coef=coef, act=act, slope=slope, C=Cout, partial=None)

    def backward():
        nonlocal x
        dz = t.pop_grad(zk)
        src = t.bn_src.pop(id(zk), None)
        if dz is None:
            return
        dres = None
        if use_bn:
            if not bn_train:
                raise NotImplementedError("backward through eval-mode BatchNorm is not supported")
            coef2 = torch.empty((2, Cout), dtype=torch.float32, device=x.device)
            dgam, acc = t.param_grad(bn.weight)
            dbet, acc2 = t.param_grad(bn.bias)
            assert acc == acc2
            dy = torch.empty_like(y)
            dres = torch.empty_like(y) if need_res else None
            part = src["partial"] if src is not None else None
            if isinstance(dz, HeadGrad) and (need_res or not _state["bn_recompute"]):
                dz = dz.materialize(t)
            if isinstance(dz, HeadGrad):
                # this layer's output feeds the one-channel output convolution and nothing else: its gradient is a function of the nine
                # neighbouring dlogits, recomputed inside the two passes of the BatchNorm backward (pass 1 also sums the head's weight gradient)
                hg, w_h = dz, dz.weight
                hrows = lib.rd_bn_head_rows(N, OH, OW)
                hpart = torch.empty((hrows, 88, 2), dtype=torch.float32, device=x.device)
                hargs = (_p(hg.dl), _p(y), _p(mean), _p(rstd), _p(scale), _p(shift), act, slope, _p(w_h.detach()))
                _chk(_timed("bn_backward", 2 * hg.flops, lambda: lib.rd_bn_head_bwd_reduce(*hargs, _p(hpart), N, OH, OW, Cout, dt, st),
                            "bn+head backward M=%d C=%d [sums + head wgrad]" % (pixels, Cout), b_out + 2 * pixels * es, kernel=_head_name(1, dt, act), idem=True),
                     "rd_bn_head_bwd_reduce")
                dwh, acch = t.param_grad(w_h) if hg.w_req else (None, 0)
                _chk(_timed("bn_backward", hg.flops, lambda: lib.rd_bn_head_bwd_apply(*hargs, _p(hpart), hrows, _p(coef2), _p(dgam), _p(dbet), acc, _p(dwh), acch,
                                                                                       _p(dy), N, OH, OW, Cout, dt, st),
                            "bn+head backward M=%d C=%d [apply + head dgrad]" % (pixels, Cout), 2 * b_out + 2 * pixels * es, kernel=_head_name(2, dt, act)),
                     "rd_bn_head_bwd_apply")
                partial = None
            elif part is not None and part[3] is dz and not need_res:
                # the data gradient that wrote this very dz tensor already summed (g, g * xhat) over it in its epilogue: finalize + apply only
                lazy_counts["bn_bwd_fused"] += 1
                _chk(_tb("bn_backward", 3 * b_out,
                         lambda: lib.rd_bn_act_bwd_from_partial(_p(dz), _p(y), _p(mean), _p(rstd), _p(scale), _p(shift), _p(part[0]), part[1], part[2],
                                                                _p(coef2), _p(dgam), _p(dbet), acc, _p(dy), pixels, Cout, act, slope, dt, st),
                         "bn backward M=%d C=%d [finalize+apply, sums from the dgrad]" % (pixels, Cout), kernel=_bn_name(2, Cout, dt, act, True)),
                     "rd_bn_act_bwd_from_partial")
                partial = None
            else:
                rows = lib.rd_bn_bwd_rows(pixels, Cout)
                partial = torch.empty((rows, Cout, 2), dtype=torch.float32, device=x.device)
            if partial is None:
                pass
            elif residual is None and _state["bn_recompute"]:   # z = act(scale*y + shift): the backward recomputes the activation argument from y, z is not read
                # algorithmic bytes: dz and y read once, dy written once (the two-pass kernels read dz and y twice)
                _chk(_bn_bwd_recompute(dz, z, y, mean, rstd, scale, shift, partial, coef2, dgam, dbet, acc, dy, dres, pixels, Cout, act, slope, dt, st,
                                       3 * b_out, "bn backward M=%d C=%d" % (pixels, Cout)), "rd_bn_act_bwd_recompute")
            else:
                _chk(_tb("bn_backward", (4 + (dres is not None)) * b_out,
                         lambda: lib.rd_bn_act_bwd(_p(dz), _p(z), _p(y), _p(mean), _p(rstd), _p(scale), _p(partial), _p(coef2), _p(dgam),
                                                   _p(dbet), acc, _p(dy), _p(dres), pixels, Cout, act, slope, dt, st),
                         "bn backward(res) M=%d C=%d" % (pixels, Cout)), "rd_bn_act_bwd")
        else:
            if isinstance(dz, HeadGrad):
                dz = dz.materialize(t)
            eff_act = act
            if eff_act != ACT_NONE:
                dy = torch.empty_like(y)
                _chk(_tb("elementwise", 3 * b_out, lambda: lib.rd_act_bwd(_p(dz), _p(z), _p(dy), dz.numel(), eff_act, slope, dt, st), "act bwd"),
                     "rd_act_bwd")
            else:
                dy = dz
            dres = dy
        if need_res:
            t.add_grad(residual, dres)
        if bias is not None and bias.requires_grad:
            db, acc = t.param_grad(bias)
            rows = lib.rd_colsum_rows(pixels, Cout)
            part = torch.empty((rows, Cout, 2), dtype=torch.float32, device=x.device)
            if _state["defer_wgrad"]:      # partial rows now, every layer's final sums in one launch at the next stage mark
                if any(b_ == id(bias) for _, _, b_ in t.colsum):      # a bias used twice: its two sums must not share a launch
                    t.flush_colsum()
                _chk(lib.rd_colsum_partial(_p(dy), _p(part), pixels, Cout, dt, st), "rd_colsum_partial")
                t.colsum.append((_lib.ColsumItem(part.data_ptr(), db.data_ptr(), rows, Cout, acc, 0), part, id(bias)))
            else:
                _chk(lib.rd_colsum(_p(dy), _p(part), _p(db), acc, pixels, Cout, dt, st), "rd_colsum")
        ve = 16 // es
        wfus = None
        if w_req and fus is not None:       # the weight gradient's x operand is the virtual z too: fused where its kernel can, else z is written now
            if _state["defer_wgrad"] and not (KH == 1 and KW == 1) and lib.rd_conv_wgrad_fusion_ok(ctypes.byref(d), ctypes.byref(fus)):
                wfus = fus
            else:
                x = lz.materialize()
        if w_req and KH == 1 and KW == 1 and stride == 1 and not is_up and _state["defer_wgrad"] and C1 % ve == 0 and Cout % ve == 0 \
                and (C2 == 0 or (C1 % 64 == 0 and C2 % ve == 0)):
            t.deferred.append(dict(x=x, x2=x2, dy=dy, weight=weight, M=pixels, C1=C1, C2=C2, Cin=Cin, Cout=Cout, flops=flops))
        elif w_req:
            dw, acc = t.param_grad(weight)
            ws = torch.empty(lib.rd_conv_wgrad_workspace_bytes(ctypes.byref(d)) // 4, dtype=torch.float32, device=x.device)
            d_real = _desc(dt, N, Hin, Win, C1_real, 0, False, Hin, Win, Cout, KH, KW, stride, pad, 1, OH, OW, conv_act, slope, Cout) if cin_pad else None
            if cin_pad and lib.rd_conv_wgrad_streams(ctypes.byref(d_real)):
                # few channels, millions of pixels: the streaming weight-gradient kernel reads the un-padded tensor directly
                ws = torch.empty(lib.rd_conv_wgrad_workspace_bytes(ctypes.byref(d_real)) // 4, dtype=torch.float32, device=x.device)
                _chk(_timed("conv_wgrad", flops, lambda: lib.rd_conv_wgrad(ctypes.byref(d_real), _p(x_real), None, _p(dy), _p(ws), _p(dw), acc, st),
                            "wgrad " + shp, b_in + b_out + weight.numel() * 4), "rd_conv_wgrad")
            elif cin_pad:   # gradient w.r.t. the zero-padded weight, then drop the padded input channels
                dwp = torch.empty((Cout, cin_pad, KH, KW), dtype=torch.float32, device=x.device)
                _chk(_timed("conv_wgrad", flops, lambda: lib.rd_conv_wgrad(ctypes.byref(d), _p(x), None, _p(dy), _p(ws), _p(dwp), 0, st),
                            "wgrad " + shp, b_in + b_out + weight.numel() * 4), "rd_conv_wgrad")
                _chk(lib.rd_unpad_weight_grad(_p(dwp), _p(dw), Cout, C1_real, cin_pad, KH * KW, acc, st), "rd_unpad_weight_grad")
            elif _state["defer_wgrad"]:      # slabs now, their reduction with every other layer's in one launch at the next stage mark
                if id(weight) in t.conv_reduce_w:      # a weight used twice: its two reductions must not share a launch
                    t.flush_conv_reduce()
                item = _lib.WgradReduceItem()
                st_w = st
                if wfus is not None:
                    lazy_counts["wgrad_fused"] += 1
                    _chk(_timed("conv_wgrad", flops, lambda: lib.rd_conv_wgrad_partial_fused(ctypes.byref(d), ctypes.byref(wfus), _p(x), _p(x2), _p(dy),
                                                                                                _p(ws), _p(dw), acc, ctypes.byref(item), st_w),
                                "wgrad " + shp + " (bn-in)", b_in + b_out + weight.numel() * 4,
                                kernel=lambda: lib.rd_conv_wgrad_fused_kernel_name(ctypes.byref(d), ctypes.byref(wfus)).decode(), idem=True), "rd_conv_wgrad_partial_fused")
                else:
                    _chk(_timed("conv_wgrad", flops, lambda: lib.rd_conv_wgrad_partial(ctypes.byref(d), _p(x), _p(x2), _p(dy), _p(ws), _p(dw), acc,
                                                                                          ctypes.byref(item), st_w),
                                "wgrad " + shp, b_in + b_out + weight.numel() * 4,
                                kernel=lambda: lib.rd_conv_wgrad_kernel_name(ctypes.byref(d)).decode(), idem=True), "rd_conv_wgrad_partial")
                t.defer_conv_reduce(item, ws, weight)
            else:
                _chk(_timed("conv_wgrad", flops, lambda: lib.rd_conv_wgrad(ctypes.byref(d), _p(x), _p(x2), _p(dy), _p(ws), _p(dw), acc, st),
                            "wgrad " + shp, b_in + b_out + weight.numel() * 4), "rd_conv_wgrad")
        if need_in:
            wpd = packed_weight(weight, 1, dt)
            Cin_d, C1_d = (C1_real, C1_real) if cin_pad else (Cin, C1)      # the data gradient has the tensor's own channels, not the padded ones
            dd = _desc(dt, N, OH, OW, Cout, 0, False, OH, OW, Cin_d, KH, KW, 1, KH - 1 - pad, stride, Hin, Win, ACT_NONE, 0.0, C1_d)
            # exact 2x nearest up-sampling (UpConv2d at 120x50 -> 240x100 ...): the kernel sums the 2x2 blocks of its output tile and stores the
            # gradient at SOURCE resolution; the full-resolution tensor and the upsample_nearest_bwd pass over it disappear
            if (is_up and C2 == 0 and (Hin, Win) == (2 * H1, 2 * W1) and KH == 3 and KW == 3 and stride == 1 and pad == 1 and not cin_pad
                    and _state["up2_dgrad"]):
                # the data gradient of the exact-2x layer ON ITS SOURCE: a 3x3 convolution of dy viewed space-to-depth (4 Cout (class, channel) channels
                # per source pixel) with the transposed per-class kernels; structurally zero K blocks skipped; no 2x2 reduction pass
                dd2 = _desc(dt, N, H1, W1, 4 * Cout, 0, False, H1, W1, C1, 3, 3, 1, 1, 1, H1, W1, ACT_NONE, 0.0, C1)
                dd2.in_s2d = 1
                if lib.rd_conv_up2_dgrad_ok(ctypes.byref(dd2)):
                    lazy_counts["up2_dgrad"] += 1
                    g1 = torch.empty_like(x)
                    wp3 = packed_weight(weight, 3, dt)
                    _chk(_timed("conv_gemm", flops * 4.0 / 9.0, lambda: lib.rd_conv_fwd(ctypes.byref(dd2), _p(dy), None, _p(wp3), None, _p(g1), None, None, st),
                                "dgrad " + shp + " (on source)", b_out + b_w + g1.numel() * es,
                                kernel=lambda: lib.rd_conv_fwd_kernel_name(ctypes.byref(dd2)).decode(), idem=True), "rd_conv_fwd(dgrad, in_s2d)")
                    t.add_grad(xk, g1)
                    return
            fused_up = False
            if is_up and C2 == 0 and (Hin, Win) == (2 * H1, 2 * W1) and _state.get("fuse_upsample_bwd", True):
                dd.out_reduce2 = 1
                fused_up = bool(lib.rd_conv_out_reduce2_ok(ctypes.byref(dd)))
                dd.out_reduce2 = 1 if fused_up else 0
            if fused_up:
                g1 = torch.empty_like(x)
                _chk(_timed("conv_gemm", flops, lambda: lib.rd_conv_fwd(ctypes.byref(dd), _p(dy), None, _p(wpd), None, _p(g1), None, None, st),
                            "dgrad " + shp + " (2x2 summed)", b_out + b_w + g1.numel() * es,
                            kernel=lambda: lib.rd_conv_fwd_kernel_name(ctypes.byref(dd)).decode(), idem=True), "rd_conv_fwd(dgrad, out_reduce2)")
                t.add_grad(xk, g1)
                return
            dxv1 = torch.empty((N, Hin, Win, C1_d), dtype=x.dtype, device=x.device)
            dxv2 = torch.empty((N, Hin, Win, C2), dtype=x.dtype, device=x.device) if C2 else None
            # x already holds a gradient contribution (a skip connection's decoder side, a residual shortcut): the kernel adds it in its
            # epilogue and the sum replaces it -- no second tensor, no separate add pass
            cur = t.grads.get(id(xk)) if (C2 == 0 and not is_up and _state.get("fuse_grad_add", True) and id(xk) in t.req) else None
            if isinstance(cur, HeadGrad):      # the one-channel head was this tensor's other consumer: its virtual gradient is written out first
                cur = t.grads[id(xk)] = cur.materialize(t)
            if cur is not None and not (cur.shape == dxv1.shape and cur.dtype == dxv1.dtype and cur.is_contiguous() and lib.rd_conv_add_ok(ctypes.byref(dd))):
                cur = None
            # x is the output of conv -> BatchNorm -> act (materialised or virtual): this launch writes its dz, so its epilogue also sums the
            # BatchNorm backward's (g, g * xhat) over what it stores -- the producer's backward then skips its reduce pass over (dz, y).  Valid
            # only if the tensor written here is the final dz (checked by identity in the producer's backward: a later contribution makes a new one).
            bsrc = t.bn_src.get(id(xk)) if not is_up and not cin_pad else None
            bfus = sb = None
            if isinstance(t.grads.get(id(xk)), HeadGrad):
                bsrc = None
            if bsrc is not None and bsrc["C"] == C1_d and id(xk) in t.req and (cur is not None or t.grads.get(id(xk)) is None):
                bfus = _lib.ConvFusion()
                cf = bsrc["coef"]
                bfus.bn_y, bfus.bn_scale, bfus.bn_shift, bfus.bn_mean, bfus.bn_rstd = _p(bsrc["y"]), _p(cf[0]), _p(cf[1]), _p(cf[2]), _p(cf[3])
                bfus.bn_act, bfus.bn_slope = bsrc["act"], bsrc["slope"]
                if lib.rd_conv_fusion_ok(ctypes.byref(dd), ctypes.byref(bfus)):
                    nrows = lib.rd_conv_stats_rows(ctypes.byref(dd))
                    sb = torch.empty((nrows, Cin_d, 2), dtype=torch.float32, device=x.device)
                else:
                    bfus = None
            if bfus is not None:
                _chk(_timed("conv_gemm", flops, lambda: lib.rd_conv_fwd_fused(ctypes.byref(dd), ctypes.byref(bfus), _p(dy), None, _p(wpd), None, _p(cur),
                                                                                 _p(dxv1), _p(dxv2), _p(sb), st),
                            "dgrad " + shp + (" (+grad, bn sums)" if cur is not None else " (bn sums)"),
                            b_out + b_w + (N * Hin * Win * Cin + N * Hin * Win * C1_d * (2 if cur is not None else 1)) * es,
                            kernel=lambda: lib.rd_conv_fused_kernel_name(ctypes.byref(dd), ctypes.byref(bfus)).decode(), idem=True), "rd_conv_fwd_fused(dgrad)")
                bsrc["partial"] = (sb, nrows, Cin_d, dxv1)
                if cur is not None:
                    t.grads[id(xk)] = dxv1
                    return
            elif cur is not None:
                _chk(_timed("conv_gemm", flops, lambda: lib.rd_conv_fwd_add(ctypes.byref(dd), _p(dy), None, _p(wpd), None, _p(cur), _p(dxv1), st),
                            "dgrad " + shp + " (+grad)", b_out + b_w + 2 * N * Hin * Win * Cin * es,
                            kernel=lambda: lib.rd_conv_fwd_kernel_name(ctypes.byref(dd)).decode(), idem=True), "rd_conv_fwd_add(dgrad)")
                t.grads[id(xk)] = dxv1
                return
            else:
                _chk(_timed("conv_gemm", flops, lambda: lib.rd_conv_fwd(ctypes.byref(dd), _p(dy), None, _p(wpd), None, _p(dxv1), _p(dxv2),
                                                                           None, st), "dgrad " + shp, b_out + b_w + N * Hin * Win * Cin * es,
                            kernel=lambda: lib.rd_conv_fwd_kernel_name(ctypes.byref(dd)).decode(), idem=True), "rd_conv_fwd(dgrad)")
            if is_up:
                g1 = torch.empty_like(x)
                _chk(_tb("elementwise", (dxv1.numel() + g1.numel()) * es,
                         lambda: lib.rd_upsample_nearest_bwd(_p(dxv1), _p(g1), N, H1, W1, Hin, Win, C1, dt, st), "upsample bwd"),
                     "rd_upsample_nearest_bwd")
                g2 = None
                if C2:
                    g2 = torch.empty_like(x2)
                    _chk(_tb("elementwise", (dxv2.numel() + g2.numel()) * es,
                             lambda: lib.rd_upsample_nearest_bwd(_p(dxv2), _p(g2), N, H1, W1, Hin, Win, C2, dt, st), "upsample bwd"),
                         "rd_upsample_nearest_bwd")
            else:
                g1, g2 = dxv1, dxv2
            t.add_grad(xk, g1)
            t.add_grad(x2, g2)

    t.record(backward)
    return zk


def linear(x, weight, *, x2=None, bias=None, act=ACT_NONE, slope=0.2):
    """Token/feature matrix (rows, C) -> (rows, Cout): nn.Linear as a 1x1 convolution over rows."""
    rows = x.shape[0]
    x4 = alias(x, x.view(rows, 1, 1, x.shape[1]))
    x24 = None if x2 is None else alias(x2, x2.view(rows, 1, 1, x2.shape[1]))
    z = conv_block(x4, weight, x2=x24, bias=bias, stride=1, pad=0, act=act, slope=slope)
    return alias(z, z.view(rows, z.shape[3]))


def alias(src, view):
    """Register `view` (a reshaped view of `src`, same memory) on the tape so gradients flow back to `src`."""
    t = tape()
    if t is not None and id(src) in t.req:
        t.mark(view)

        def backward():
            g = t.pop_grad(view)
            if g is not None:
                t.add_grad(src, g.reshape(src.shape))
        t.record(backward)
    return view


# ------------------------------------------------------------------------------------------- pooling
def maxpool(x, k=3, s=2, p=1):
    lib, t, dt, st = L(), tape(), rd_of(x), _stream(x)
    N, H, W, C = x.shape
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    out = torch.empty((N, OH, OW, C), dtype=x.dtype, device=x.device)
    arg = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device)
    _chk(_tb("pooling", x.numel() * x.element_size() + out.numel() * (x.element_size() + 1),
             lambda: lib.rd_maxpool_fwd(_p(x), _p(out), _p(arg), N, H, W, C, OH, OW, k, s, p, dt, st), "maxpool fwd"), "rd_maxpool_fwd")
    if t is not None and t.requires(x):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is None:
                return
            dx = torch.empty_like(x)
            _chk(_tb("pooling", x.numel() * x.element_size() + out.numel() * (x.element_size() + 1),
                     lambda: lib.rd_maxpool_bwd(_p(g), _p(arg), _p(dx), N, H, W, C, OH, OW, k, s, p, dt, st), "maxpool bwd"), "rd_maxpool_bwd")
            t.add_grad(x, dx)
        t.record(backward)
    return out


_roi_flags = {}      # device -> scratch flag of forwards WITHOUT a backward (inference, validation): written by the kernel, read by nobody
_roi_live = []       # the flags of the most recent training forwards (one per tape), kept for check_roi_overflow()


def _roi_flag(x, t):
    """Overflow flag of the compact RoI arg-max: raised by a forward that met a bin window it cannot encode (> 15 pixels in a direction);
    the backward kernels of THAT forward then write NaN gradients -- a wrong geometry cannot pass silently.  The flag is scoped to the
    training forward (one zeroed word per tape, shared by its five poolings); a forward without a backward gets a per-device scratch word
    that no backward ever reads, so an inference or validation pass can never poison a later training step."""
    dev = x.device
    if t is None or not t.requires(x):
        f = _roi_flags.get(dev)
        if f is None:
            f = _roi_flags[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
        return f
    f = getattr(t, "roi_flag", None)
    if f is None or f.device != dev:
        f = t.roi_flag = torch.zeros(1, dtype=torch.int32, device=dev)      # a fill kernel (also under graph capture: re-zeroed by every replay)
        _roi_live.append(f)
        del _roi_live[:-8]
    return f


def check_roi_overflow():
    """Host-side check of the compact RoI arg-max (synchronises): raises if one of the recent training forwards (the last eight tapes, incl. the
    one a captured step replays) met a bin window beyond 15 pixels -- its RoI-pool gradients are NaN.  Called where the host waits anyway
    (FlatAdam.state_dict, RCNetModel.save_model, the end of bench.py's timed region); a training loop may call it whenever it reads the loss."""
    live = [f for f in _roi_live]
    if not live:
        return
    bad = [int(v) for v in torch.stack([f.reshape(()) for f in live]).cpu().tolist()]
    if any(bad):
        _roi_live[:] = [f for f, b in zip(live, bad) if not b]
        raise RuntimeError("riders_amd: roi_pool met a bin window wider than 15 pixels, which the one-byte arg-max cannot encode; the gradients of "
                           "that step are NaN.  Use engine.set_roi_u8(False) (int32 arg-max) for this geometry.")


def set_roi_u8(flag):
    """Compact one-byte RoI arg-max (default) or int32 pixel indices (any geometry)."""
    set_switch("roi_u8", bool(flag))


def roi_argmax(out):
    """Arg-max of an engine.roi_pool output as int32 pixel indices h * W + w (-1 = empty bin), torchvision's convention; the compact
    one-byte form is decoded on the host with the forward's window arithmetic (tests / debugging; the kernels never need the indices)."""
    arg = out._rd_argmax
    if arg.dtype == torch.int32:
        return arg
    import numpy as np
    rois, (H, W, PH, PW, scale) = out._rd_roi_geom
    a = arg.cpu().numpy().astype(np.int64)
    r = rois.detach().cpu().numpy().astype(np.float32)
    f32 = np.float32

    def starts(lo, hi, P, lim):      # clamp(floor(p * bin) + start), all in float32 as the kernels compute it
        rnd = lambda v: (np.sign(v) * np.floor(np.abs(v) + f32(0.5))).astype(np.int64)      # roundf: halves away from zero
        s, e = rnd(lo * f32(scale)), rnd(hi * f32(scale))
        ext = np.maximum(e - s + 1, 1)
        binsz = (ext.astype(np.float32) / f32(P))[:, None]
        return np.clip(np.floor(np.arange(P, dtype=np.float32)[None, :] * binsz).astype(np.int64) + s[:, None], 0, lim)
    hs, ws = starts(r[:, 2], r[:, 4], PH, H), starts(r[:, 1], r[:, 3], PW, W)
    idx = (hs[:, :, None, None] + (a >> 4)) * W + ws[:, None, :, None] + (a & 15)
    return torch.from_numpy(np.where(a == 255, -1, idx).astype(np.int32))


def roi_pool(x, rois, output_size, spatial_scale, compact=None, out=None):
    """x (N,H,W,C); rois (R,5) fp32 = (batch idx, x1, y1, x2, y2) -> (R,PH,PW,C); the arg-max is kept for the backward as int32 pixel indices
    or (compact, the default where the kernels of this shape support it: round 4) as ONE BYTE per element, the arg-max's offset inside its bin
    window -- `roi_argmax(out)` gives the indices either way."""
    lib, t, dt, st = L(), tape(), rd_of(x), _stream(x)
    N, H, W, C = x.shape
    R = rois.shape[0]
    PH, PW = int(output_size[0]), int(output_size[1])
    if out is None:
        out = torch.empty((R, PH, PW, C), dtype=x.dtype, device=x.device)
    else:      # written in place, e.g. into its half of the transformer's token matrix (no copy behind the pooling)
        assert tuple(out.shape) == (R, PH, PW, C) and out.dtype == x.dtype and out.is_contiguous()
    if compact is None:
        compact = _state["roi_u8"]
    compact = bool(compact) and C % (16 // x.element_size()) == 0 and _state["roi_bwd"] in ("auto", "gather", "atomic") and PH < (1 << 19) and PW < (1 << 19) \
        and _state["roi_tile_min_blocks"] > 0      # (the LDS tile-accumulate backward, an A/B form, reads int32 indices)
    arg = torch.empty((R, PH, PW, C), dtype=torch.uint8 if compact else torch.int32, device=x.device)
    flag = _roi_flag(x, t) if compact else None
    # algorithmic bytes: pooled values + argmax written, the source map read once
    roi_bytes = out.numel() * (x.element_size() + arg.element_size()) + x.numel() * x.element_size()
    if compact:
        _chk(_tb("roi_pool", roi_bytes, lambda: lib.rd_roi_pool_fwd_u8(_p(x), _p(rois), _p(out), _p(arg), _p(flag), R, N, H, W, C, PH, PW,
                                                                        float(spatial_scale), dt, st), "roi_pool fwd %dx%d C=%d" % (PH, PW, C)), "rd_roi_pool_fwd_u8")
    else:
        _chk(_tb("roi_pool", roi_bytes, lambda: lib.rd_roi_pool_fwd(_p(x), _p(rois), _p(out), _p(arg), R, N, H, W, C, PH, PW, float(spatial_scale), dt, st),
                 "roi_pool fwd %dx%d C=%d" % (PH, PW, C)), "rd_roi_pool_fwd")
    out._rd_argmax = arg
    out._rd_roi_geom = (rois, (H, W, PH, PW, float(spatial_scale)))
    if t is not None and t.requires(x):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is None:
                return
            nblk = ((H + 15) // 16) * ((W + 15) // 16) * N * ((C + 31) // 32)
            big = nblk >= max(_state["roi_tile_min_blocks"], 1)
            gather = _state["deterministic_roi_pool"] or _state["roi_bwd"] == "gather" or (_state["roi_bwd"] == "auto" and big and _state["roi_tile_min_blocks"] > 0)
            if compact:      # pixel-owner gather on the large maps (or everywhere when determinism is asked for), fp32 L2 atomics on the small ones
                if gather:
                    dx = torch.empty_like(x)
                    _chk(_tb("roi_pool", roi_bytes, lambda: lib.rd_roi_pool_bwd_gather_u8(_p(g), _p(rois), _p(arg), _p(flag), _p(dx), R, N, H, W, C, PH, PW,
                                                                                             float(spatial_scale), dt, st),
                             "roi_pool bwd(gather) %dx%d C=%d" % (PH, PW, C)), "rd_roi_pool_bwd_gather_u8")
                    t.add_grad(x, dx)
                    return
                dx32 = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
                _chk(_tb("roi_pool", roi_bytes, lambda: lib.rd_roi_pool_bwd_u8(_p(g), _p(rois), _p(arg), _p(flag), _p(dx32), R, N, H, W, C, PH, PW,
                                                                                float(spatial_scale), dt, st),
                         "roi_pool bwd(atomic) %dx%d C=%d" % (PH, PW, C)), "rd_roi_pool_bwd_u8")
                t.add_grad(x, cast(dx32, x.dtype))
                return
            if gather and C % (16 // x.element_size()) == 0:   # pixel-owner gather: fixed summation order, no atomics
                dx = torch.empty_like(x)
                _chk(_tb("roi_pool", roi_bytes, lambda: lib.rd_roi_pool_bwd_gather(_p(g), _p(rois), _p(arg), _p(dx), R, N, H, W, C, PH, PW,
                                                                                      float(spatial_scale), dt, st),
                         "roi_pool bwd(gather) %dx%d C=%d" % (PH, PW, C)), "rd_roi_pool_bwd_gather")
                t.add_grad(x, dx)
                return
            if C % 32 == 0 and H * W < (1 << 24) and _state["roi_bwd"] in ("auto", "tile") and nblk >= _state["roi_tile_min_blocks"]:   # LDS tile accumulators
                dx = torch.empty_like(x)
                _chk(_tb("roi_pool", roi_bytes, lambda: lib.rd_roi_pool_bwd_tile(_p(g), _p(rois), _p(arg), _p(dx), R, N, H, W, C, PH, PW,
                                                                                    float(spatial_scale), dt, st),
                         "roi_pool bwd(tile) %dx%d C=%d" % (PH, PW, C)), "rd_roi_pool_bwd_tile")
                t.add_grad(x, dx)
                return
            dx32 = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
            _chk(_tb("roi_pool", roi_bytes, lambda: lib.rd_roi_pool_bwd(_p(g), _p(rois), _p(arg), _p(dx32), R, N, H, W, C, PH, PW, dt, st),
                     "roi_pool bwd(atomic) %dx%d C=%d" % (PH, PW, C)), "rd_roi_pool_bwd")
            t.add_grad(x, cast(dx32, x.dtype))
        t.record(backward)
    return out


# ------------------------------------------------------------------------------------- token-level ops
def layernorm(x, ln, residual=None):
    """(rows, C) -> residual + LayerNorm(x) with ln.weight / ln.bias (torch.nn.LayerNorm as container)."""
    lib, t, dt, st = L(), tape(), rd_of(x), _stream(x)
    rows, C = x.shape
    out = torch.empty_like(x)
    stat = torch.empty((2, rows), dtype=torch.float32, device=x.device)
    _chk(lib.rd_layernorm_fwd(_p(x), _p(ln.weight.detach()), _p(ln.bias.detach()), _p(residual), _p(out), _p(stat[0]), _p(stat[1]),
                              rows, C, float(ln.eps), dt, st), "rd_layernorm_fwd")
    if t is not None and (t.requires(x, residual) or ln.weight.requires_grad):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is None:
                return
            dx = torch.empty_like(x)
            nb = lib.rd_layernorm_bwd_rows(rows)
            part = torch.empty((nb, C, 2), dtype=torch.float32, device=x.device)
            dg, acc = t.param_grad(ln.weight)
            db, acc2 = t.param_grad(ln.bias)
            _chk(lib.rd_layernorm_bwd(_p(g), _p(x), _p(ln.weight.detach()), _p(stat[0]), _p(stat[1]), _p(dx), _p(part), _p(dg), _p(db),
                                      acc, rows, C, dt, st), "rd_layernorm_bwd")
            t.add_grad(x, dx)
            t.add_grad(residual, g)
        t.record(backward)
    return out


def linear_attention(q, k, v, N, Lq, S, H, eps=1e-6):
    """q (N*L, H*16), k/v (N*S, H*16) token matrices -> (N*L, H*16)."""
    lib, t, dt, st = L(), tape(), rd_of(q), _stream(q)
    C = q.shape[1]
    assert C == H * 16, "linear_attention kernel is specialised for head dim 16"
    out = torch.empty_like(q)
    _chk(lib.rd_linear_attention_fwd(_p(q), _p(k), _p(v), _p(out), N, Lq, S, H, C, C, C, C, eps, dt, st), "rd_linear_attention_fwd")
    if t is not None and t.requires(q, k, v):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is None:
                return
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            _chk(lib.rd_linear_attention_bwd(_p(q), _p(k), _p(v), _p(g), _p(dq), _p(dk), _p(dv), N, Lq, S, H, C, C, C, C, eps, dt, st),
                 "rd_linear_attention_bwd")
            t.add_grad(q, dq)
            t.add_grad(k, dk)
            t.add_grad(v, dv)
        t.record(backward)
    return out


def _common_base(a, b):
    """the contiguous 2-D matrix whose two row ranges a and b are (in either order), or None"""
    base = getattr(a, "_base", None)
    if base is None or base is not getattr(b, "_base", None) or base.dim() != 2 or a.dim() != 2 or b.dim() != 2:
        return None
    if not (base.is_contiguous() and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype == base.dtype):
        return None
    if a.shape[1] != base.shape[1] or b.shape[1] != base.shape[1] or a.shape[0] + b.shape[0] != base.shape[0]:
        return None
    es, p0 = base.element_size(), base.data_ptr()
    lo, hi = (a, b) if a.data_ptr() <= b.data_ptr() else (b, a)
    if lo.data_ptr() != p0 or hi.data_ptr() != p0 + lo.numel() * es:
        return None
    return base


def rows_cat(a, b):
    """(Ra, C), (Rb, C) -> one (Ra + Rb, C) token matrix (copy).  The LoFTR transformer keeps both token streams in ONE buffer so that a
    'self' layer -- the same weights on both streams, no interaction (RCNet/linear_attention.py:171-173) -- is one launch over all
    2 N sequences instead of two launches of N workgroups each on a 256-CU chip."""
    t = tape()
    ra, rb = a.shape[0], b.shape[0]
    base = _common_base(a, b)
    if base is not None and a.data_ptr() == base.data_ptr():      # a and b ARE the two row ranges of one matrix (their producers wrote in place)
        return rows_join(a, b, base)
    out = torch.empty((ra + rb, a.shape[1]), dtype=a.dtype, device=a.device)
    for src, dst in ((a, out[:ra]), (b, out[ra:])):
        _chk(L_().rd_cast(_p(src), _p(dst), src.numel(), rd_of(src), rd_of(dst), 1.0, _stream(src)), "rd_cast")
    if t is not None and t.requires(a, b):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is not None:
                t.add_grad(a, g[:ra])
                t.add_grad(b, g[ra:])
        t.record(backward)
    return out


def rows_split(x, ra):
    """(R, C) -> views (x[:ra], x[ra:]); the backward gathers the two gradients into one matrix (they come from different launches)."""
    t = tape()
    a, b = x[:ra], x[ra:]
    if t is not None and t.requires(x):
        t.mark(a); t.mark(b)

        def backward():
            ga, gb = t.pop_grad(a), t.pop_grad(b)
            if ga is None and gb is None:
                return
            base = getattr(ga, "_base", None) if ga is not None else None
            if base is not None and gb is not None and getattr(gb, "_base", None) is base and base.shape == x.shape and base.dtype == x.dtype \
                    and base.is_contiguous() and ga.data_ptr() == base.data_ptr() and gb.data_ptr() == base.data_ptr() + ga.numel() * ga.element_size():
                t.add_grad(x, base)     # the two gradients ARE the halves of one matrix (CrossGrad): no copy
                return
            g = torch.empty_like(x)
            for src, dst in ((ga, g[:ra]), (gb, g[ra:])):
                if src is None:
                    dst.zero_()
                else:
                    _chk(L_().rd_cast(_p(src), _p(dst), src.numel(), rd_of(src), rd_of(dst), 1.0, _stream(src)), "rd_cast")
            t.add_grad(x, g)
        t.record(backward)
    return a, b


def rows_join(a, b, whole):
    """a and b are the two row ranges of `whole` (already written in place by their producers): returns `whole` as a tape tensor whose
    gradient flows back to a and b as views (no copy)."""
    t = tape()
    if t is not None and t.requires(a, b):
        t.mark(whole)
        ra = a.shape[0]

        def backward():
            g = t.pop_grad(whole)
            if g is not None:
                t.add_grad(a, g[:ra])
                t.add_grad(b, g[ra:])
        t.record(backward)
    return whole


class CrossGrad(object):
    """Shared by the two cross-attention calls of one transformer layer (a2 = layer(a, b); b2 = layer(b, a2)) so that their backward
    launches write the gradient of the layer's input matrix [a; b] in place: call 2 stores d b into the lower half and ADDS its source
    gradient to a2's pending gradient (rd_loftr_grads.dsrc_accumulate); call 1 stores d a into the upper half and adds its source gradient
    to the lower one.  Without it every cross layer costs two gradient-add passes and two row copies (16 launches per RC-Net step)."""

    def __init__(self):
        self.G = None


def loftr_layer(x, source, layer, N, L, S, out=None, cross=None, cross_role=0):
    """Fused LoFTREncoderLayer (rd_loftr_layer_fwd / _bwd): x (N*L, 128), source (N*S, 128) token matrices -> (N*L, 128)
    (written into `out` when given: a row range of a larger token matrix).
    `layer` is the nn.Module holding q_proj / k_proj / v_proj / merge / mlp / norm1 / norm2 (reference linear_attention.py:84-135).
    The weight gradients of the six linears join the tape's grouped weight-gradient launch."""
    lib, t, dt, st = L_(), tape(), rd_of(x), _stream(x)
    C = x.shape[1]
    same = source is x
    ws = [layer.q_proj.weight, layer.k_proj.weight, layer.v_proj.weight, layer.merge.weight, layer.mlp[0].weight, layer.mlp[2].weight]
    lns = [layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias]

    def wstruct(mode):
        w = _lib.LoftrWeights()
        packs = [packed_weight(p, mode, dt) for p in ws]
        w.wq, w.wk, w.wv, w.wm, w.w0, w.w2 = [b.data_ptr() for b in packs]
        w.g1, w.b1, w.g2, w.b2 = [p.detach().data_ptr() for p in lns]
        return w, packs
    ML, MS = N * L, N * S
    dev, T = x.device, x.dtype
    q = torch.empty((ML, C), dtype=T, device=dev)
    k, v = torch.empty((MS, C), dtype=T, device=dev), torch.empty((MS, C), dtype=T, device=dev)
    att, mpre, msg, m2pre = (torch.empty((ML, C), dtype=T, device=dev) for _ in range(4))
    hid = torch.empty((ML, 2 * C), dtype=T, device=dev)
    stats = torch.empty((ML, 4), dtype=torch.float32, device=dev)
    if out is None:
        out = torch.empty_like(x)
    assert out.shape == x.shape and out.dtype == x.dtype and out.is_contiguous()
    sv = _lib.LoftrSaved()
    sv.q, sv.k, sv.v, sv.att, sv.mpre, sv.msg, sv.hid, sv.m2pre, sv.stats = [b.data_ptr() for b in (q, k, v, att, mpre, msg, hid, m2pre, stats)]
    wf, keep = wstruct(0)
    eps_a, eps_l = float(layer.attention.eps), float(layer.norm1.eps)
    flops = 2.0 * (ML * C * C * 2 + MS * C * C * 2 + ML * 2 * C * 2 * C + ML * 2 * C * C)   # the six token GEMMs
    _chk(_timed("loftr_layer", flops, lambda: lib.rd_loftr_layer_fwd(_p(x), _p(source), ctypes.byref(wf), _p(out), ctypes.byref(sv), N, L, S,
                                                                       eps_a, eps_l, dt, st), "fwd N=%d L=%d" % (N, L)), "rd_loftr_layer_fwd")
    if t is None or not (t.requires(x, source) or any(p.requires_grad for p in ws + lns)):
        return out
    t.mark(out)
    saved_keep = (q, k, v, att, mpre, msg, hid, m2pre, stats)   # `sv` holds raw pointers: the tensors must outlive the backward

    def backward():
        g = t.pop_grad(out)
        if g is None:
            return
        assert len(saved_keep) == 9
        if not g.is_contiguous():
            g = g.contiguous()
        dm2pre, dmpre, datt, dq, dx = (torch.empty((ML, C), dtype=T, device=dev) for _ in range(5))
        dk, dv = torch.empty((MS, C), dtype=T, device=dev), torch.empty((MS, C), dtype=T, device=dev)
        dhid = torch.empty((ML, 2 * C), dtype=T, device=dev)
        dsrc = None if same else torch.empty((MS, C), dtype=T, device=dev)
        # cross-attention pair writing the gradient of [a; b] in place (CrossGrad): role 2 = the second forward call (x = b, source = a2),
        # whose backward runs first; role 1 = the first call (x = a, source = b)
        in_place = 0
        if cross is not None and not same and ML == MS and _state.get("loftr_cross_inplace", True):
            if cross_role == 2:
                pend = t.grads.get(id(source))
                if pend is not None and pend.shape == (MS, C) and pend.dtype == T and pend.is_contiguous():
                    cross.G = torch.empty((ML + MS, C), dtype=T, device=dev)
                    dx, dsrc, in_place = cross.G[ML:], pend, 2
            elif cross_role == 1 and cross.G is not None:
                dx, dsrc, in_place = cross.G[:ML], cross.G[ML:], 1
        lnp = torch.empty((2, N, C, 2), dtype=torch.float32, device=dev)
        (dg1, a1), (db1, a2), (dg2, a3), (db2, a4) = [t.param_grad(p) for p in lns]
        assert a1 == a2 == a3 == a4
        gr = _lib.LoftrGrads()
        gr.dout, gr.dm2pre, gr.dhid, gr.dmpre, gr.datt, gr.dq, gr.dk, gr.dv, gr.dx = [b.data_ptr() for b in (g, dm2pre, dhid, dmpre, datt, dq, dk, dv, dx)]
        gr.dsrc = 0 if dsrc is None else dsrc.data_ptr()
        gr.lnp1, gr.lnp2 = lnp[0].data_ptr(), lnp[1].data_ptr()
        gr.dg1, gr.db1, gr.dg2, gr.db2, gr.accumulate = dg1.data_ptr(), db1.data_ptr(), dg2.data_ptr(), db2.data_ptr(), a1
        gr.defer_ln = 1 if _state["defer_wgrad"] else 0      # the LayerNorm partials of every application of a stage are summed by ONE launch
        gr.dsrc_accumulate = 1 if in_place else 0
        wb, keepb = wstruct(1)
        _chk(_timed("loftr_layer", 2.0 * flops, lambda: lib.rd_loftr_layer_bwd(_p(x), _p(source), ctypes.byref(wb), ctypes.byref(sv), ctypes.byref(gr),
                                                                                 N, L, S, eps_a, dt, st), "bwd N=%d L=%d" % (N, L)), "rd_loftr_layer_bwd")
        if gr.defer_ln:
            t.defer_ln_grad(lns[0], dg1, db1, a1, lnp[0], N)
            t.defer_ln_grad(lns[2], dg2, db2, a1, lnp[1], N)
        for w_, x1, x2, dy, M in ((ws[0], x, None, dq, ML), (ws[1], source, None, dk, MS), (ws[2], source, None, dv, MS),
                                  (ws[3], att, None, dmpre, ML), (ws[4], x, msg, dhid, ML), (ws[5], hid, None, dm2pre, ML)):
            if w_.requires_grad:
                c1 = x1.shape[1]
                c2 = 0 if x2 is None else x2.shape[1]
                t.deferred.append(dict(x=x1, x2=x2, dy=dy, weight=w_, M=M, C1=c1, C2=c2, Cin=c1 + c2, Cout=dy.shape[1],
                                       flops=2.0 * M * (c1 + c2) * dy.shape[1]))
        if in_place == 2:
            return                      # d b waits in cross.G for call 1; a2's pending gradient was updated in place
        t.add_grad(x, dx)
        if not same:
            t.add_grad(source, dsrc)    # (in_place == 1: the two halves of cross.G, which rows_split's backward recognises as one matrix)
        if in_place == 1:
            cross.G = None
    t.record(backward)
    return out


def transpose_last2(x, B, R, Cc):
    """contiguous [B][R][Cc] -> [B][Cc][R] (any leading shape; returns a flat (B, Cc, R) tensor)."""
    lib, t, dt, st = L(), tape(), rd_of(x), _stream(x)
    out = torch.empty((B, Cc, R), dtype=x.dtype, device=x.device)
    _chk(lib.rd_transpose_last2(_p(x), _p(out), B, R, Cc, dt, st), "rd_transpose_last2")
    if t is not None and t.requires(x):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is None:
                return
            dx = torch.empty_like(x)
            _chk(lib.rd_transpose_last2(_p(g), _p(dx), B, Cc, R, dt, st), "rd_transpose_last2")
            t.add_grad(x, dx)
        t.record(backward)
    return out


def concat_channels(a, b):
    """(..., Ca), (..., Cb) -> (..., Ca+Cb) on the innermost (channel) axis."""
    lib, t, dt, st = L(), tape(), rd_of(a), _stream(a)
    Ca, Cb = a.shape[-1], b.shape[-1]
    rows = a.numel() // Ca
    out = torch.empty(a.shape[:-1] + (Ca + Cb,), dtype=a.dtype, device=a.device)
    _chk(lib.rd_concat2(_p(a), _p(b), _p(out), rows, Ca, Cb, dt, st), "rd_concat2")
    if t is not None and t.requires(a, b):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is None:
                return
            base = _common_base(a, b) if a.dim() == 2 else None
            if base is not None:      # a and b are the halves of one token matrix: so are their gradients (rows_split's backward then needs no copy)
                Gm = torch.empty_like(base)
                ra_ = a.shape[0]
                ga, gb = (Gm[:ra_], Gm[ra_:]) if a.data_ptr() == base.data_ptr() else (Gm[b.shape[0]:], Gm[:b.shape[0]])
            else:
                ga, gb = torch.empty_like(a), torch.empty_like(b)
            _chk(lib.rd_split2(_p(g), _p(ga), _p(gb), rows, Ca, Cb, dt, st), "rd_split2")
            t.add_grad(a, ga)
            t.add_grad(b, gb)
        t.record(backward)
    return out


def input_cast(x, scale=1.0, out=None):
    """Region input (fp32) -> activation dtype; gradient flows back as fp32.  out: written there (a row range of a larger token matrix)."""
    t = tape()
    if out is not None:
        assert out.shape == x.shape and out.is_contiguous() and x.is_contiguous()
        _chk(L().rd_cast(_p(x), _p(out), x.numel(), rd_of(x), rd_of(out), float(scale), _stream(x)), "rd_cast")
    else:
        out = to_act(x, scale)
    if out is x:
        return x
    if t is not None and t.requires(x):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is not None:
                t.add_grad(x, cast(g, x.dtype, scale))
        t.record(backward)
    return out


def output_cast(x, dtype):
    t = tape()
    out = cast(x, dtype)
    if out is x:
        return x
    if t is not None and t.requires(x):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is not None:
                t.add_grad(x, cast(g, x.dtype))
        t.record(backward)
    return out


def add_act(a, b, act=ACT_NONE, slope=0.2):
    """act(a + b) on same-shape NHWC tensors (ResNet block tail, utils/net_utils.py:323)."""
    ak, bk = a, b               # tape keys of the operands (a LazyAct stays the key of its z, also after it has been materialised)
    if isinstance(a, LazyAct):  # act(act1(scale*y + shift) + b) in one pass over y and b: z is never written
        lz = a
        if isinstance(b, LazyAct):
            b = b.materialize()
        y = lz.y
        dt, st, C = rd_of(y), _stream(y), y.shape[-1]
        if L().rd_affine_act_add_ok(C, dt):
            lib, t = L(), tape()
            lazy_counts["add_fused"] += 1
            out = torch.empty_like(y)
            _chk(_tb("elementwise", 3 * y.numel() * y.element_size(),
                     lambda: lib.rd_affine_act_add(_p(y), _p(lz.coef[0]), _p(lz.coef[1]), lz.act, lz.slope, _p(b), _p(out), y.numel() // C, C, act, slope,
                                                   dt, st), "bn apply+act+add+act"), "rd_affine_act_add")
            a = None
        else:
            a = lz.materialize()
    b = materialize(b)
    if a is not None:
        lib, t, dt, st = L(), tape(), rd_of(a), _stream(a)
        C = a.shape[-1]
        out = torch.empty_like(a)
        _chk(_tb("elementwise", 3 * a.numel() * a.element_size(),
                 lambda: lib.rd_affine_act(_p(a), None, None, _p(b), _p(out), a.numel() // C, C, act, slope, dt, st), "add+act"), "rd_affine_act")
    if t is not None and t.requires(ak, bk):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is None:
                return
            if act != ACT_NONE:
                d = torch.empty_like(g)
                _chk(lib.rd_act_bwd(_p(g), _p(out), _p(d), g.numel(), act, slope, dt, st), "rd_act_bwd")
            else:
                d = g
            t.add_grad(ak, d)
            t.add_grad(bk, d)
        t.record(backward)
    return out


def from_nchw(x, scale=1.0, dtype=None):
    """Region input, logical NCHW (contiguous or channels_last, fp32 or act dtype) -> (N,H,W,C) engine tensor."""
    t = tape()
    out = nchw_to_nhwc(x, scale, dtype)
    if t is not None and t.requires(x):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is None:
                return
            gi = torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device)
            n, c, h, w = x.shape
            if x.permute(0, 2, 3, 1).is_contiguous():
                _chk(L().rd_cast(_p(g), _p(gi), g.numel(), rd_of(g), rd_of(gi), float(scale), _stream(g)), "rd_cast")
            else:
                assert scale == 1.0
                _chk(L().rd_nhwc_to_nchw(_p(g), _p(gi), n, c, h, w, rd_of(g), rd_of(gi), _stream(g)), "rd_nhwc_to_nchw")
            t.add_grad(x, gi)
        t.record(backward)
    return out


def to_nchw_out(z, dtype=None):
    """(N,H,W,C) engine tensor -> logical NCHW region output (channels_last view, optionally cast)."""
    t = tape()
    zc = z if dtype is None else cast(z, dtype)
    out = zc.permute(0, 3, 1, 2)
    if t is not None and t.requires(z):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is not None:
                t.add_grad(z, cast(g.permute(0, 2, 3, 1), z.dtype))
        t.record(backward)
    return out


def tokens_in(x, C):
    """Region input [..., C-ish] (any leading shape) -> contiguous (rows, C) token matrix in the activation dtype."""
    xc = x if x.is_contiguous() else x.contiguous()
    v = alias(x, xc.view(-1, C))
    return input_cast(v)


def alias_cast_view(o, dtype, shape):
    """(rows, C) engine tensor -> region output of `dtype` viewed as `shape` (same element order)."""
    t = tape()
    out = cast(o, dtype).view(shape)
    if t is not None and t.requires(o):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is not None:
                gc = g if g.is_contiguous() else g.contiguous()
                t.add_grad(o, cast(gc.view(o.shape), o.dtype))
        t.record(backward)
    return out


def sigmoid(x):
    out = torch.empty_like(x)
    _chk(L().rd_sigmoid(_p(x), _p(out), x.numel(), rd_of(x), _stream(x)), "rd_sigmoid")
    return out


# ----------------------------------------------------------------------------------------- RC-Net loss
def rcnet_labels(ground_truth_depth, radar_points, max_distance, all_valid=False):
    """RCNet/rcnet_main.py:308-332: gt (R,1,H,W) fp32, points (R,3) -> (label, validity) fp32 tensors."""
    R = ground_truth_depth.shape[0]
    hw = ground_truth_depth.numel() // max(R, 1)
    gt = ground_truth_depth if ground_truth_depth.is_contiguous() else ground_truth_depth.contiguous()
    pts = radar_points if radar_points.is_contiguous() else radar_points.contiguous()
    label, valid = torch.empty_like(gt), torch.empty_like(gt)
    _chk(L().rd_rcnet_labels(_p(gt), _p(pts), _p(label), _p(valid), R, hw, float(max_distance), 1 if all_valid else 0,
                             _stream(gt)), "rd_rcnet_labels")
    return label, valid


def bce_masked(logits, label, valid, pos_weight):
    """sum(valid * BCEWithLogits(logits, label, pos_weight)) / sum(valid) as a 0-dim fp32 tensor."""
    lib, t, st = L(), tape(), _stream(logits)
    lg = logits if logits.is_contiguous() else logits.contiguous()  # (R,1,H,W): C == 1, NCHW == NHWC
    lg = alias(logits, lg) if lg is not logits else lg
    n = lg.numel()
    rows = lib.rd_bce_rows(n)
    partial = torch.empty((rows, 2), dtype=torch.float32, device=lg.device)
    out = torch.empty(3, dtype=torch.float32, device=lg.device)  # [loss, sum valid*bce, sum valid]
    loss, sums = out[0:1], out[1:3]
    _chk(lib.rd_bce_masked_fwd(_p(lg), _p(label), _p(valid), float(pos_weight), _p(partial), _p(loss), _p(sums), n, rd_of(lg), st),
         "rd_bce_masked_fwd")
    res = out[0]
    if t is not None and t.requires(lg):
        t.mark(res)

        def backward():
            g = t.pop_grad(res)
            if g is None:
                return
            g32 = g if g.dtype == torch.float32 else cast(g.reshape(1), torch.float32)
            d = torch.empty_like(lg)
            _chk(lib.rd_bce_masked_bwd(_p(lg), _p(label), _p(valid), float(pos_weight), _p(sums), _p(g32), _p(d), n, rd_of(lg), st),
                 "rd_bce_masked_bwd")
            t.add_grad(lg, d)
        t.record(backward)
    return res


# =================================================================================== Scale Map Learner ops
def _bn_forward(y, bn, act, slope, residual, training, stats=None):
    """BatchNorm (+residual, +activation) on an NHWC tensor whose producer has no fused statistics epilogue."""
    lib, dt, st = L(), rd_of(y), _stream(y)
    C = y.shape[-1]
    pixels = y.numel() // C
    bn_train = training or not bn.track_running_stats
    if bn_train and stats is None:
        rows = lib.rd_dw_rows(pixels, C)
        stats = torch.empty((rows, C, 2), dtype=torch.float32, device=y.device)
        _chk(lib.rd_bn_stats(_p(y), _p(stats), pixels, C, dt, st), "rd_bn_stats")
    coef = torch.empty((4, C), dtype=torch.float32, device=y.device)
    if bn_train and residual is None and lib.rd_bn_slab_ok(pixels, C, dt):
        z = torch.empty_like(y)
        _chk(_bn_finalize_apply(stats, y, bn, coef, z, pixels, C, act, slope, dt, st), "rd_bn_finalize_apply")
        return z, coef, bn_train
    _chk(lib.rd_bn_finalize(_p(stats), 0 if stats is None else stats.shape[0], C, float(pixels), _p(bn.weight.detach()), _p(bn.bias.detach()),
                            float(bn.eps), float(bn.momentum if bn.momentum is not None else 0.1), 1 if bn_train else 0,
                            _p(bn.running_mean), _p(bn.running_var), _p(coef[2]), _p(coef[3]), _p(coef[0]), _p(coef[1]), st), "rd_bn_finalize")
    z = torch.empty_like(y)
    _chk(_tb("bn_apply", (2 + (residual is not None)) * y.numel() * y.element_size(),
             lambda: lib.rd_affine_act(_p(y), _p(coef[0]), _p(coef[1]), _p(residual), _p(z), pixels, C, act, slope, dt, st),
             "bn apply+act M=%d C=%d" % (pixels, C), kernel=_bn_name(0, C, dt, act, residual is not None)), "rd_affine_act")
    return z, coef, bn_train


def _bn_backward(t, dz, z, y, coef, bn, act, slope, want_res):
    lib, dt, st = L(), rd_of(y), _stream(y)
    C = y.shape[-1]
    pixels = y.numel() // C
    rows = lib.rd_bn_bwd_rows(pixels, C)
    partial = torch.empty((rows, C, 2), dtype=torch.float32, device=y.device)
    coef2 = torch.empty((2, C), dtype=torch.float32, device=y.device)
    dgam, acc = t.param_grad(bn.weight)
    dbet, _ = t.param_grad(bn.bias)
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_res else None
    if not _state["bn_recompute"]:
        _chk(lib.rd_bn_act_bwd(_p(dz), _p(z), _p(y), _p(coef[2]), _p(coef[3]), _p(coef[0]), _p(partial), _p(coef2), _p(dgam), _p(dbet), acc,
                               _p(dy), _p(dres), pixels, C, act, slope, dt, st), "rd_bn_act_bwd")
        return dy, dres
    # no residual on this path: the activation argument is recomputed from y (coef[0] = scale, coef[1] = shift), z is not read
    _chk(_bn_bwd_recompute(dz, z, y, coef[2], coef[3], coef[0], coef[1], partial, coef2, dgam, dbet, acc, dy, dres, pixels, C, act, slope, dt, st,
                           3 * y.numel() * y.element_size(), "bn backward M=%d C=%d" % (pixels, C)), "rd_bn_act_bwd_recompute")
    return dy, dres




def dwconv_block(x, weight, *, stride=1, pad=0, out_hw=None, bn=None, act=ACT_NONE, slope=0.0, training=True):
    """act(BN(depthwise_conv(x))) for the EfficientNet-Lite blocks; weight (C,1,k,k) fp32."""
    lib, t, dt, st = L(), tape(), rd_of(x), _stream(x)
    N, H, W, C = x.shape
    k = weight.shape[-1]
    OH, OW = (int(out_hw[0]), int(out_hw[1])) if out_hw is not None else ((H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1)
    y = torch.empty((N, OH, OW, C), dtype=x.dtype, device=x.device)
    stats = None
    srows = lib.rd_dwconv_stats_rows(N, OH, OW, C, k, stride) if (_state["dw_fused_stats"] and bn is not None and (training or not bn.track_running_stats)) else 0
    if srows > 0:      # the BatchNorm statistics come out of the convolution's epilogue: no separate pass over y
        stats = torch.empty((srows, C, 2), dtype=torch.float32, device=x.device)
        _chk(lib.rd_dwconv_fwd_stats(_p(x), _p(weight.detach()), _p(y), _p(stats), N, H, W, C, OH, OW, k, stride, pad, dt, st), "rd_dwconv_fwd_stats")
    else:
        _chk(lib.rd_dwconv_fwd(_p(x), _p(weight.detach()), _p(y), N, H, W, C, OH, OW, k, stride, pad, dt, st), "rd_dwconv_fwd")
    if bn is not None:
        z, coef, bn_train = _bn_forward(y, bn, act, slope, None, training, stats=stats)
    else:
        z, coef, bn_train = y, None, False
        assert act == ACT_NONE
    if t is None or not (t.requires(x) or weight.requires_grad):
        return z
    t.mark(z)

    def backward():
        dz = t.pop_grad(z)
        if dz is None:
            return
        if bn is not None:
            if not bn_train:
                raise NotImplementedError("backward through eval-mode BatchNorm is not supported")
            dy, _ = _bn_backward(t, dz, z, y, coef, bn, act, slope, False)
        else:
            dy = dz
        if weight.requires_grad:
            dw, acc = t.param_grad(weight)
            rows = lib.rd_dw_rows(N * OH * OW, C)
            part = torch.empty((rows, C, k * k), dtype=torch.float32, device=x.device)
            if _state["defer_wgrad"]:      # partial rows now, every depthwise layer's final sums in one launch at the next stage mark
                if any(w_ == id(weight) for _, _, w_ in t.dw_reduce):
                    t.flush_dw_reduce()
                item = _lib.DwWgradItem()
                _chk(lib.rd_dwconv_wgrad_partial(_p(x), _p(dy), _p(part), _p(dw), acc, N, H, W, C, OH, OW, k, stride, pad, dt, ctypes.byref(item), st),
                     "rd_dwconv_wgrad_partial")
                if item.rows > 0:
                    t.dw_reduce.append((item, part, id(weight)))
            else:
                _chk(lib.rd_dwconv_wgrad(_p(x), _p(dy), _p(part), _p(dw), acc, N, H, W, C, OH, OW, k, stride, pad, dt, st), "rd_dwconv_wgrad")
        if t.requires(x):
            dx = torch.empty_like(x)
            _chk(lib.rd_dwconv_dgrad(_p(dy), _p(weight.detach()), _p(dx), N, H, W, C, OH, OW, k, stride, pad, dt, st), "rd_dwconv_dgrad")
            t.add_grad(x, dx)
    t.record(backward)
    return z


def activation(x, act, slope=0.0):
    """Stand-alone activation (pre-activation ReLU of the residual conv units, modules/midas/blocks.py:107)."""
    lib, t, dt, st = L(), tape(), rd_of(x), _stream(x)
    C = x.shape[-1]
    out = torch.empty_like(x)
    _chk(lib.rd_affine_act(_p(x), None, None, None, _p(out), x.numel() // C, C, act, slope, dt, st), "rd_affine_act")
    if t is not None and t.requires(x):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is None:
                return
            d = torch.empty_like(g)
            _chk(lib.rd_act_bwd(_p(g), _p(out), _p(d), g.numel(), act, slope, dt, st), "rd_act_bwd")
            t.add_grad(x, d)
        t.record(backward)
    return out


def bilinear2x(x, align_corners):
    """F.interpolate(scale_factor=2, mode='bilinear') on NHWC (modules/midas/blocks.py:168-170, :187)."""
    lib, t, dt, st = L(), tape(), rd_of(x), _stream(x)
    N, H, W, C = x.shape
    OH, OW, al = 2 * H, 2 * W, 1 if align_corners else 0
    out = torch.empty((N, OH, OW, C), dtype=x.dtype, device=x.device)
    _chk(lib.rd_bilinear_fwd(_p(x), _p(out), N, H, W, C, OH, OW, al, dt, st), "rd_bilinear_fwd")
    if t is not None and t.requires(x):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is None:
                return
            dx = torch.empty_like(x)
            _chk(lib.rd_bilinear_bwd(_p(g), _p(dx), N, H, W, C, OH, OW, al, dt, st), "rd_bilinear_bwd")
            t.add_grad(x, dx)
        t.record(backward)
    return out


def sml_head(out, d, min_pred, max_pred):
    """pred = d * relu(1 + out) with the in-place clamps of midas_net_custom.py:121-133; out (N,H,W,1), d fp32 same size."""
    lib, t, dt, st = L(), tape(), rd_of(out), _stream(out)
    hi = 1.0 / min_pred if min_pred is not None else -1.0
    lo = 1.0 / max_pred if max_pred is not None else -1.0
    pred = torch.empty(out.shape, dtype=torch.float32, device=out.device)
    n = out.numel()
    _chk(lib.rd_sml_head_fwd(_p(out), _p(d), _p(pred), n, hi, lo, dt, st), "rd_sml_head_fwd")
    if t is not None and t.requires(out):
        t.mark(pred)

        def backward():
            g = t.pop_grad(pred)
            if g is None:
                return
            do = torch.empty_like(out)
            _chk(lib.rd_sml_head_bwd(_p(out), _p(d), _p(g), _p(do), n, hi, lo, dt, st), "rd_sml_head_bwd")
            t.add_grad(out, do)
        t.record(backward)
    return pred


def reciprocal(x):
    lib, t, st = L(), tape(), _stream(x)
    out = torch.empty_like(x)
    _chk(lib.rd_reciprocal(_p(x), None, _p(out), x.numel(), st), "rd_reciprocal")
    if t is not None and t.requires(x):
        t.mark(out)

        def backward():
            g = t.pop_grad(out)
            if g is None:
                return
            dx = torch.empty_like(x)
            _chk(lib.rd_reciprocal(_p(x), _p(g), _p(dx), x.numel(), st), "rd_reciprocal")
            t.add_grad(x, dx)
        t.record(backward)
    return out


def sml_loss(pred, image, gt_interp, gt_sparse, weights, w_lidar, w_smooth, w_edge, filter_size, loss_kind=0):
    """utils/loss.py compute_loss on (N,1,H,W) fp32 contiguous tensors -> (info[7] tensor).  loss_kind 0 'l1', 1 'l2', 2 'smoothl1' (:55-100);
    w_edge > 0 adds the edge-matching term (:241-249) to the loss AND to the saved gradient fields (needs w_smooth > 0)."""
    lib, t, st = L(), tape(), _stream(pred)
    N, _, H, W = pred.shape
    n = pred.numel()
    rows = lib.rd_sml_loss_rows(n)
    partial = torch.empty((rows, 8), dtype=torch.float64, device=pred.device)
    gfx, gfy = torch.empty_like(pred), torch.empty_like(pred)
    info = torch.empty(7, dtype=torch.float32, device=pred.device)
    _chk(lib.rd_sml_loss_fwd_kind(_p(pred), _p(image), _p(gt_interp), _p(gt_sparse), _p(weights), N, H, W, filter_size, int(loss_kind), w_lidar, w_smooth,
                                  w_edge, _p(gfx), _p(gfy), _p(partial), _p(info), st), "rd_sml_loss_fwd_kind")
    loss = info[0]
    if t is not None and t.requires(pred):
        t.mark(loss)

        def backward():
            g = t.pop_grad(loss)
            if g is None:
                return
            dp = torch.empty_like(pred)
            _chk(lib.rd_sml_loss_bwd_kind(_p(pred), _p(gt_interp), _p(gt_sparse), _p(gfx), _p(gfy), _p(info), _p(g), N, H, W, filter_size, int(loss_kind),
                                          w_lidar, w_smooth, _p(dp), st), "rd_sml_loss_bwd_kind")
            t.add_grad(pred, dp)
        t.record(backward)
    return loss, info


def outlier_removal(depth, kernel_size, threshold):
    lib, st = L(), _stream(depth)
    N, _, H, W = depth.shape
    part = torch.empty(lib.rd_outlier_parts(depth.numel()), dtype=torch.float32, device=depth.device)
    out = torch.empty_like(depth)
    _chk(lib.rd_outlier_removal(_p(depth), _p(part), _p(out), N, H, W, int(kernel_size), float(threshold), st), "rd_outlier_removal")
    return out
