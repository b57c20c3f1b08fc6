"""SML loss on MI355X; same signature and return structure as the reference's utils/loss.py compute_loss :5-135
(train_zju.py:459-470 configures 'l1', w_edge = 0, w_unsupervised = 0, single-scale output; 'l2', 'smoothl1' and w_edge > 0 run on the same
kernels since round 6; w_unsupervised -- a median over a boolean mask -- does not)."""
import torch

from . import engine


def compute_loss(image, output_depth, gt_interp, gt_sparse, loss_func, w_smoothness, sobel_filter_size, validity_map_loss_smoothness,
                 w_lidar_loss, w_edge, invalid_map_gt, w_unsupervised):
    kinds = {'l1': 0, 'l2': 1, 'smoothl1': 2}
    if loss_func not in kinds:
        raise ValueError('No such loss: {}'.format(loss_func))      # utils/loss.py:103
    if w_edge > 0.0 and not w_smoothness > 0.0:
        raise NotImplementedError("w_edge > 0 needs w_smoothness > 0 here (the saved gradient fields are stored per unit of w_smoothness)")
    if w_unsupervised > 0.0:
        raise NotImplementedError("w_unsupervised > 0 is not used (train_zju.py:470)")
    if isinstance(output_depth, (list, tuple)):
        if len(output_depth) != 1:
            raise NotImplementedError("multi-scale outputs are not produced by MidasNet_small_videpth")
        output_depth = output_depth[0]
    if image.shape[1] != 1:
        raise NotImplementedError("the SML loop passes the 1-channel depth as `image` (train_zju.py:374-376)")
    c = lambda a: a if a.is_contiguous() else a.contiguous()  # noqa: E731
    img, gi, gs = c(image.float()), c(gt_interp.float()), c(gt_sparse.float())
    wts = None if validity_map_loss_smoothness is None else c(validity_map_loss_smoothness.float())

    def run(pred):
        p = pred if pred.is_contiguous() else engine.alias(pred, pred.contiguous())
        loss, info = engine.sml_loss(p, img, gi, gs, wts, float(w_lidar_loss), float(w_smoothness), float(w_edge), int(sobel_filter_size), kinds[loss_func])
        run.info = info
        return loss
    loss = engine.run_region(run, (output_depth,), [])
    info = run.info
    loss_info = {'loss': loss, 'loss_supervised': info[1], 'loss_lidar': info[2] if w_lidar_loss > 0 else 0.0,
                 'loss_smoothness': info[3], 'loss_edge': info[4], 'loss_unsupervised': 0.0}
    return loss, loss_info
