"""Scale-Map-Learner parity on a real MI355X (HIP kernels through the C ABI) vs the oracle and the reference's fixtures."""
import pytest

from tests import parity_cases_sml as S

pytestmark = pytest.mark.gpu


def test_effnet_blocks(gpu):
    S.effnet_block_case(gpu, "ir", 24, 32, 3, 2)
    S.effnet_block_case(gpu, "ir", 16, 16, 5, 1, H=7, W=9)
    S.effnet_block_case(gpu, "ds", 32, 24, 3, 1, H=8, W=8)
    S.effnet_block_case(gpu, "ir", 136, 232, 5, 2, H=18, W=24)
    S.effnet_block_case(gpu, "ir", 48, 48, 5, 1, H=19, W=30)      # sliding-window depthwise kernels, ragged last run
    S.effnet_block_case(gpu, "ir", 24, 24, 3, 1, H=10, W=13)


def test_bilinear(gpu):
    S.bilinear_case(gpu)


def test_fusion_block(gpu):
    S.fusion_block_case(gpu)
    S.fusion_block_case(gpu, f=256)


def test_loss_and_outlier(gpu):
    S.loss_case(gpu)


def test_prestep(gpu):
    S.prestep_case(gpu)


def test_prestep_scale_and_shift(gpu):
    S.prestep_st_case(gpu)


def test_metrics(gpu):
    S.metrics_case(gpu)


def test_sml_network_golden(gpu):
    S.sml_net_case(gpu)


def test_bf16_depthwise_exact(gpu):
    """bf16 depthwise kernels (sliding-window forward / data gradient, run-based weight gradient) bit-exact on integer data."""
    S.bf16_exact_dwconv_case(gpu)
    S.bf16_exact_dwconv_case(gpu, C=144, k=3, s=2, H=18, W=24)
    S.bf16_exact_dwconv_case(gpu, C=32, k=3, s=1, H=9, W=7, N=1)
    S.bf16_exact_dwconv_case(gpu, C=232 * 6, k=5, s=2, H=18, W=24, N=2)
    S.bf16_exact_dwconv_case(gpu, C=24 * 6, k=3, s=2, H=36, W=48, N=2)


def test_sml_bf16_blocks(gpu):
    """configs[2] precision mode: EfficientNet-Lite3 blocks and a fusion block in bf16 (the `*_gen` BatchNorm passes, depthwise
    kernels and 1x1 direct kernels the throughput numbers run on) against the oracle rounding at the same tensors
    (oracle/precision.py).  Relative L2: forward / running statistics 2e-3, input and parameter gradients 6e-3 (measured 4e-5 .. 2.7e-3);
    fusion block max-norm 1e-2 (= 2.5 bf16 ulp)."""
    with S.bf16_mode():
        S.effnet_block_case(gpu, "ir", 24, 32, 3, 2, H=24, W=32, N=4, tol=2e-3, l2=True)
        S.effnet_block_case(gpu, "ds", 32, 24, 3, 1, H=16, W=16, N=4, tol=2e-3, l2=True)
        S.effnet_block_case(gpu, "ir", 136, 232, 5, 2, H=18, W=24, N=4, tol=2e-3, l2=True)
        S.effnet_block_case(gpu, "ir", 48, 48, 5, 1, H=19, W=30, N=4, tol=2e-3, l2=True)
        S.fusion_block_case(gpu, f=64, tol=1e-2)


def test_sml_network_bf16(gpu):
    S.sml_net_bf16_case(gpu)


def test_sml_config2_full_size(gpu):
    S.sml_config2_fullsize_case(gpu)


def test_sml_bf16_trains_like_fp32(gpu):
    S.sml_bf16_convergence_case(gpu)


def test_validation_chain_abs_rel(gpu):
    S.validate_chain_case(gpu)


def test_sml_train_step_and_validate(gpu):
    """End-to-end SML step (device pre-step -> net -> 1/pred -> outlier removal -> loss -> backward -> fused Adam) decreases the loss,
    and validation returns finite metrics."""
    import numpy as np
    import torch
    from riders_amd import sml_main
    from riders_amd.optim import FlatAdam
    torch.manual_seed(0)
    model = sml_main.build_model(gpu)
    model.train()
    opt = FlatAdam(model.parameters(), lr=1e-4)
    batch = sml_main.synthetic_batch(2, 96, 128, seed=3, device=gpu)
    orr = sml_main.make_outlier_removal()
    losses = [float(sml_main.train_step(model, opt, batch, outlier=orr)) for _ in range(6)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    model.eval()
    r = sml_main.validate_batch(model, batch)
    assert np.isfinite(r["abs_rel"]).all() and (r["count"] > 0).all()


def test_sml_full_size_backward(gpu):
    S.sml_fullsize_backward_case(gpu)


def test_sml_config3_per_rank_share(gpu):
    S.sml_config3_rank_case(gpu)


def test_sml_config4_per_rank_batch(gpu):
    S.sml_config4_rank_case(gpu)
