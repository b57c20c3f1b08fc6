"""RC-Net training / inference harness on MI355X: the counterpart of the reference's RCNet/rcnet_main.py for the
hot path (train step body :272-359, forward_output :435-487) plus the synthetic ZJU-shape batch generator that
restates the tensor contract of data/datasets.py:170-288.  File I/O, TensorBoard and the
augmentation transforms are out of scope.
"""
import ctypes
import os

import torch
import torch.distributed

from . import engine
from .networks import boxes_to_rois
from .rcnet_model import RCNetModel

# RCNet/train_rcnet_zju.py:28-65
ZJU_CONFIG = dict(
    patch_size=[240, 100], total_points_sampled=30, batch_size=4,
    input_channels_image=3, input_channels_depth=3,
    encoder_type=['rcnet', 'batch_norm'], n_filters_encoder_image=[32, 64, 128, 128, 128],
    n_neurons_encoder_depth=[32, 64, 128, 128, 128],
    decoder_type=['multiscale', 'batch_norm'], n_filters_decoder=[256, 128, 64, 32, 16],
    weight_initializer='kaiming_uniform', activation_func='leaky_relu',
    learning_rate=2e-4, w_positive_class=2.5, max_distance_correspondence=0.5, set_invalid_to_negative_class=False,
)


def build_model(device, cfg=ZJU_CONFIG):
    return RCNetModel(cfg['input_channels_image'], cfg['input_channels_depth'], cfg['patch_size'], cfg['encoder_type'],
                      cfg['n_filters_encoder_image'], cfg['n_neurons_encoder_depth'], cfg['decoder_type'],
                      cfg['n_filters_decoder'], cfg['weight_initializer'], cfg['activation_func'], device=device)


def synthetic_batch(batch_size, height=256, width=512, cfg=ZJU_CONFIG, seed=1234, device='cpu'):
    """The tuple RCNetTrainingDataset.__getitem__ returns, batched (data/datasets.py:288):
    image (B,3,H+ph,W+pw) fp32 0..255 edge-padded; radar_points (B,K,3) = (x+pad_x, y+pad_y, z);
    bounding_boxes (B,K,4) = (x-pad_x, y-pad_y, x+pad_x, y+pad_y) in padded coords; ground truth crops
    (B,K,1,ph,pw) cut from a zero-padded synthetic dense depth with ~50% invalid pixels."""
    g = torch.Generator().manual_seed(seed)
    ph, pw = cfg['patch_size']
    pad_y, pad_x = ph // 2, pw // 2
    K = cfg['total_points_sampled']
    B = batch_size
    img = torch.randint(0, 256, (B, 3, height, width), generator=g).float()
    img = torch.nn.functional.pad(img, (pad_x, pad_x, pad_y, pad_y), mode='replicate')
    x = torch.randint(0, width, (B, K), generator=g).float()
    y = torch.randint(0, height, (B, K), generator=g).float()
    z = torch.rand((B, K), generator=g) * 98.5 + 1.5
    pts = torch.stack([x + pad_x, y + pad_y, z], dim=-1)
    boxes = torch.stack([pts[..., 0] - pad_x, pts[..., 1] - pad_y, pts[..., 0] + pad_x, pts[..., 1] + pad_y], dim=-1)
    depth = torch.rand((B, 1, height, width), generator=g) * 98.5 + 1.5
    depth = depth * (torch.rand((B, 1, height, width), generator=g) > 0.5).float()
    # make a share of pixels agree with the radar depth so that positives exist
    depth = torch.nn.functional.pad(depth, (pad_x, pad_x, pad_y, pad_y))
    crops = torch.empty((B, K, 1, ph, pw))
    for b in range(B):
        for k in range(K):
            xs, ys = int(pts[b, k, 0]) - pad_x, int(pts[b, k, 1]) - pad_y
            c = depth[b, :, ys:ys + ph, xs:xs + pw].clone()
            m = (torch.rand(c.shape, generator=g) < 0.05) & (c > 0)
            c[m] = z[b, k] + 0.25
            crops[b, k] = c
    return tuple(t.to(device) for t in (img, pts, boxes, crops))


def prepare_batch(batch):
    """Device-side batch preparation of the train loop (rcnet_main.py:283-340): /255 normalise (rcnet_transforms.py:258),
    flatten points and crops, per-image box list -> ROI rows."""
    image, radar_point, boxes, gt = batch
    image = engine.as_nchw(engine.nchw_to_nhwc(image, scale=1.0 / 255.0))
    B, K = radar_point.shape[0], radar_point.shape[1]
    radar_point = radar_point.reshape(B * K, radar_point.shape[2])
    gt = gt.reshape(B * K, gt.shape[2], gt.shape[3], gt.shape[4])
    rois = boxes_to_rois(boxes)
    return image, radar_point, rois, gt


def scaled_backward(loss, optimizer, loss_scale):
    """loss.backward() with the static loss scale of the fp16 mode (the seed gradient is the scale; Adam divides it out again)."""
    optimizer.loss_scale = float(loss_scale)
    if loss_scale == 1.0:
        loss.backward()
    else:
        loss.backward(torch.full_like(loss, float(loss_scale)))


def train_step(model, optimizer, batch, cfg=ZJU_CONFIG, reducer=None, loss_scale=1.0):
    """One optimisation step (rcnet_main.py:294-359).  Returns the loss as a device tensor (no host sync)."""
    loss = forward_loss(model, batch, cfg)
    optimizer.zero_grad()
    scaled_backward(loss, optimizer, loss_scale)
    if reducer is not None:
        reducer.reduce()
    optimizer.step()
    return loss


def _scatter(crops, pts, ph, pw, H, W, response_thr):
    depth = torch.empty((1, H, W), dtype=torch.float32, device=crops.device)
    resp = torch.empty((1, H, W), dtype=torch.float32, device=crops.device)
    engine._chk(engine.L().rd_scatter_crops(engine._p(crops), engine._p(pts), engine._p(depth), engine._p(resp), crops.shape[0], ph, pw,
                                            H, W, ctypes.c_float(response_thr), engine.rd_of(crops), engine._stream(crops)),
                "rd_scatter_crops")
    return depth, resp


def forward_crops(model, image, radar_points, bounding_boxes_list):
    """First half of rcnet_main.py:435-487: edge-pad, one forward over all N points -> (crops (N,1,ph,pw) sigmoid responses, points)."""
    ph, pw = model.input_patch_size_image
    pad_y, pad_x = ph // 2, pw // 2
    image = torch.nn.functional.pad(image, (pad_x, pad_x, pad_y, pad_y), mode='replicate')  # data prep, as the reference
    if radar_points.dim() == 3:
        radar_points = torch.squeeze(radar_points, dim=0)
    pts = radar_points.contiguous().float()
    crops = model.forward(image=image, point=pts, bounding_boxes=bounding_boxes_list, return_logits=False)
    return crops, pts, (image.shape[-2] - 2 * pad_y, image.shape[-1] - 2 * pad_x)


def forward_output(model, image, radar_points, bounding_boxes_list, response_thr=0.5, device=None):
    """rcnet_main.py:435-487: edge-pad, one forward over all N points, threshold, integer paste, confidence-weighted
    mean depth; zero where nothing responds.  image (1,3,H,W) already normalised; radar_points (N,3) in padded coords."""
    ph, pw = model.input_patch_size_image
    crops, pts, (H, W) = forward_crops(model, image, radar_points, bounding_boxes_list)
    return _scatter(crops, pts, ph, pw, H, W, response_thr)


def points_to_boxes(radar_points, patch_size):
    """run_rcnet_zju.py:221-234: (N,3) radar points in image coordinates -> (points shifted into the padded image, RoI rows
    (0, x - pad_x, y - pad_y, x + pad_x, y + pad_y)); one launch instead of the reference's per-point Python loop."""
    pad_x, pad_y = patch_size[1] // 2, patch_size[0] // 2
    pts = radar_points.reshape(-1, radar_points.shape[-1]).contiguous().float()
    N = pts.shape[0]
    out = torch.empty_like(pts)
    rois = torch.empty((N, 5), dtype=torch.float32, device=pts.device)
    engine._chk(engine.L().rd_points_to_rois(engine._p(pts), engine._p(out), engine._p(rois), N, float(pad_x), float(pad_y), 0,
                                             engine._stream(pts)), "rd_points_to_rois")
    return out, rois


def fuse_with_retry(crops, pts, ph, pw, H, W, response_thr=0.5, thr_step=0.05):
    """run_rcnet_zju.py:250-264: fuse the crops (rcnet_main.py:460-485); while the fused depth sums to zero lower the threshold by 0.05."""
    thr = response_thr
    total = torch.empty(1, dtype=torch.float64, device=crops.device)
    while True:
        depth, resp = _scatter(crops, pts, ph, pw, H, W, thr)
        engine._chk(engine.L().rd_sum_f32(engine._p(depth), depth.numel(), engine._p(total), engine._stream(depth)), "rd_sum_f32")
        if float(total) != 0.0:
            return depth, resp, thr
        if thr < -1.0:
            raise RuntimeError("fuse_with_retry: no radar point responds at any threshold (no points?)")
        thr = thr - thr_step


def run_frame(model, image, radar_points, response_thr=0.5, thr_step=0.05):
    """One frame of run_rcnet_zju.py:204-266: box construction (:221-234), /255 normalisation (transforms, :236-240), forward_output
    (:242-248) and the retry loop (:250-264) that lowers the response threshold by 0.05 while the fused depth map is all zeros.
    The reference re-runs the whole network per retry; the crops do not depend on the threshold, so only the scatter is repeated here.
    image (1,3,H,W) 0..255, radar_points (N,3) or (1,N,3) image coordinates.  Returns (depth (H,W) fp32, response (H,W), threshold)."""
    ph, pw = model.input_patch_size_image
    pts, rois = points_to_boxes(radar_points, (ph, pw))
    img = engine.as_nchw(engine.nchw_to_nhwc(image.float(), scale=1.0 / 255.0, dtype=torch.float32))
    with torch.no_grad():
        crops, pts, (H, W) = forward_crops(model, img, pts, rois)
        depth, resp, thr = fuse_with_retry(crops, pts, ph, pw, H, W, response_thr, thr_step)
    return depth[0], resp[0], thr


def forward_loss(model, batch, cfg=ZJU_CONFIG):
    """Label build + forward + masked BCE of one step (rcnet_main.py:294-352) -> loss tensor."""
    image, radar_point, rois, gt = prepare_batch(batch)
    label, valid = engine.rcnet_labels(gt, radar_point, cfg['max_distance_correspondence'], cfg['set_invalid_to_negative_class'])
    logits = model.forward(image, radar_point, rois, return_logits=True)
    loss, _ = model.compute_loss(logits=logits, ground_truth=label, validity_map=valid, w_positive_class=cfg['w_positive_class'])
    return loss


def compute_gradients(model, optimizer, batch, cfg=ZJU_CONFIG, loss_scale=1.0):
    """Forward + loss + backward of one step (everything of rcnet_main.py:294-358 except the optimizer) through torch.autograd,
    as an unchanged training script would run it."""
    loss = forward_loss(model, batch, cfg)
    optimizer.zero_grad()
    scaled_backward(loss, optimizer, loss_scale)
    return loss


def staged_gradients(fwd_loss, optimizer, on_stage=None, loss_scale=1.0):
    """The same step on ONE engine tape driven from this thread (engine.StepTape, no torch.autograd): the backward stops at every
    stage mark and calls `on_stage(tag)`, which is where GraphedStep ends one hipGraph capture and begins the next.  The stage hooks
    registered with the engine (the all-reducer's) fire at the same marks."""
    optimizer.zero_grad()
    optimizer.loss_scale = float(loss_scale)
    st = engine.StepTape()
    loss = st.forward(fwd_loss)
    st.seed(loss, loss_scale)
    while True:
        tag = st.backward_stage()
        if tag is None:
            break
        if on_stage is not None:
            on_stage(tag)
    st.finish()
    return loss


class GraphedStep(object):
    """The launch-bound part of a step (forward + loss + backward: ~1100 small kernel launches for RC-Net) captured once into
    hipGraphs and replayed per step; the RCCL gradient all-reduce and the fused Adam launch stay eager.  With an all-reducer the capture
    is SPLIT at the model's stage marks: after the graph of a stage has been enqueued the all-reducer starts that stage's bucket on the
    communication stream, so the exchange overlaps the replay of the remaining backward graphs.  Without one it is a single graph.
    `fwd_loss()` runs forward + loss on static device tensors and returns the loss tensor; `on_replay(n)` keeps host-side bookkeeping
    (BatchNorm num_batches_tracked counters) in step with replays; `buffers` are tensors the forward updates in place (BatchNorm running
    statistics): the warm-up passes needed before a capture are undone on them, so constructing a GraphedStep trains nothing."""

    def __init__(self, fwd_loss, optimizer, reducer=None, warmup=2, on_replay=None, buffers=(), loss_scale=1.0):
        self.opt, self.reducer, self.on_replay = optimizer, reducer, on_replay
        assert engine._timer["t"] is None, "kernel timing and graph capture are exclusive"
        buffers = list(buffers)
        hooks, engine._stage_hooks[:] = list(engine._stage_hooks), []      # no collectives during warm-up / capture
        try:
            saved = [b.clone() for b in buffers]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    staged_gradients(fwd_loss, optimizer, None, loss_scale)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            with torch.no_grad():
                for b, s_ in zip(buffers, saved):
                    b.copy_(s_)
            self.graphs, self.tags = [], []
            pool = torch.cuda.graph_pool_handle()
            state = {}
            # The library's own collectives (parallel.RcclComm: rd_allreduce_bucket on its communication stream) are stream operations and
            # are CAPTURED: forward + backward + every bucket's exchange + the final join are one graph, the fork / join of the communication
            # stream are graph edges (round 5; replaces the per-stage graphs below for that transport).
            self.captured_comm = reducer is not None and getattr(reducer, "comm", None) is not None
            if self.captured_comm:      # RCCL sets up channels / buffers lazily on a communicator's first collective: not inside a capture
                warm = torch.zeros(256, dtype=torch.float32, device=optimizer.flat_grad.device)
                reducer.comm.all_reduce(warm)
                reducer.comm.join(warm)
                torch.cuda.synchronize()
            # With an all-reducer the capture is SPLIT at the stage marks (one graph per stage; measured 5 % slower than one graph at one rank
            # before any wire time).  ONE graph with an external event recorded at each mark (an event-record node the communication side could
            # wait on) was built in round 4 and cannot run: torch-rocm 2.10 raises "External events are disallowed in rocm" for
            # torch.cuda.Event(external=True) under capture, and an event recorded behind torch's back on a capturing stream is not something
            # that can be validated on a 1-GPU pool (a mis-ordered wait would all-reduce unfinished gradients silently).
            def begin():
                state["g"] = torch.cuda.CUDAGraph()
                # thread_local whenever a communicator lives in this process: RCCL's proxy thread and torch.distributed's NCCL watchdog thread
                # (event queries) touch the runtime while this thread captures; in the default global mode such a call invalidates the
                # capture or aborts the process (seen once in ~15 runs of the one-rank RCCL test in round 4, and again in round 5)
                relaxed = reducer is not None or (torch.distributed.is_available() and torch.distributed.is_initialized())
                state["ctx"] = torch.cuda.graph(state["g"], pool=pool, **({"capture_error_mode": "thread_local"} if relaxed else {}))
                state["ctx"].__enter__()

            def end(tag):
                state["ctx"].__exit__(None, None, None)
                self.graphs.append(state["g"])
                self.tags.append(tag)

            def boundary(tag):
                end(tag)
                begin()

            begin()
            try:
                # without an all-reducer there is nothing to interleave: one graph (each extra graph launch costs ~1 % of an RC-Net step)
                if self.captured_comm:
                    self.loss = staged_gradients(fwd_loss, optimizer, reducer.on_stage, loss_scale)
                    reducer.reduce()      # the remaining buckets + the join, inside the capture
                else:
                    self.loss = staged_gradients(fwd_loss, optimizer, boundary if reducer is not None else None, loss_scale)
            except BaseException:
                state["ctx"].__exit__(None, None, None)
                raise
            end(None)
            # a replay writes the gradient arena without passing the optimizer's gradient allocator: remember what the capture pass touched
            self.touched = optimizer.touched_indices() if hasattr(optimizer, "touched_indices") else None
            if self.touched is not None and not self.touched:
                raise RuntimeError("GraphedStep: the captured backward wrote no parameter gradient")
        finally:
            engine._stage_hooks[:] = hooks
        if on_replay is not None:
            on_replay(-(warmup + 1))  # neither the warm-up passes (undone) nor the capture pass (not executed) were training steps

    def load_batch(self, batch):
        """Copy a new batch into the static tensors the captured step reads (same shapes / dtypes; device-to-device or host-to-device copies on the
        current stream, ordered before the next replay).  This is the one line a training loop adds in front of `step()` when it feeds real data:
        `step.load_batch(next(loader)); loss = step()`."""
        static = getattr(self, "static_batch", None)
        if static is None:
            raise RuntimeError("this GraphedStep was built from a closure, not from a batch: there are no static input tensors to load into")
        if len(static) != len(batch):
            raise ValueError("load_batch: %d tensors, the captured step reads %d" % (len(batch), len(static)))
        for dst, src in zip(static, batch):
            if tuple(dst.shape) != tuple(src.shape):
                raise ValueError("load_batch: shape %s does not match the captured %s (a hipGraph replays fixed shapes)" % (tuple(src.shape), tuple(dst.shape)))
            if src.is_cuda:
                engine.dev_copy(dst, src)      # a kernel, not a hipMemcpyAsync, in front of the replay (engine.dev_copy)
            else:
                dst.copy_(src, non_blocking=True)

    def __call__(self):
        for g, tag in zip(self.graphs, self.tags):
            g.replay()
            if tag is not None and self.reducer is not None:
                self.reducer.on_stage(tag)
        if self.on_replay is not None:
            self.on_replay(+1)
        if self.reducer is not None and not self.captured_comm:
            self.reducer.reduce()
        if self.touched is not None:      # the replay filled these slots, whatever zero_grad() calls happened since the capture
            self.opt.mark_touched(self.touched)
        self.opt.step()
        return self.loss


class GraphedTrainStep(GraphedStep):
    """RC-Net training step with forward+backward replayed from hipGraphs (inputs: the static tensors of `batch`)."""

    def __init__(self, model, optimizer, batch, cfg=ZJU_CONFIG, reducer=None, warmup=2, loss_scale=1.0):
        nets = [RCNetModel._unwrap(model.encoder), RCNetModel._unwrap(model.decoder)]
        bn = [m for net in nets for m in net.modules() if getattr(m, 'use_batch_norm', False) and hasattr(m, '_nbt_pending')]

        def bump(delta):
            for m in bn:
                if m.training:
                    m._nbt_pending += delta
        buffers = [b for net in nets for b in net.buffers() if b.is_floating_point()]
        self.static_batch = tuple(batch)
        super().__init__(lambda: forward_loss(model, batch, cfg), optimizer, reducer, warmup, bump, buffers, loss_scale)
