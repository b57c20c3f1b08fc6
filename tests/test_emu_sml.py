"""SML kernel/host logic under the fiber emulator (see tests/test_emu_ops.py for what these are and are not)."""
from tests import parity_cases_sml as S


def test_effnet_blocks(emu):
    S.effnet_block_case(emu, "ir", 24, 32, 3, 2)
    S.effnet_block_case(emu, "ir", 16, 16, 5, 1, H=7, W=9)
    S.effnet_block_case(emu, "ds", 32, 24, 3, 1, H=8, W=8)


def test_depthwise_quad_kernels_exact(emu):
    """Every template of the sliding-window depthwise family (k 3/5, stride 1/2, both pad parities of the stride-2 data gradient, ragged
    last run, channel counts that are not a power of two) on integer data, bit for bit against torch, bf16 and fp32."""
    import torch
    for dtype in (torch.bfloat16, torch.float32):
        S.bf16_exact_dwconv_case(emu, C=24, k=5, s=1, H=6, W=7, N=1, dtype=dtype)
        S.bf16_exact_dwconv_case(emu, C=12, k=3, s=1, H=5, W=9, N=2, dtype=dtype)
        S.bf16_exact_dwconv_case(emu, C=20, k=3, s=2, H=8, W=10, N=1, dtype=dtype)     # even size: pad 0
        S.bf16_exact_dwconv_case(emu, C=20, k=5, s=2, H=8, W=10, N=1, dtype=dtype)     # even size: pad 1
        S.bf16_exact_dwconv_case(emu, C=8, k=3, s=2, H=9, W=11, N=1, dtype=dtype)      # odd size: pad 1
        S.bf16_exact_dwconv_case(emu, C=8, k=5, s=2, H=9, W=11, N=2, dtype=dtype)      # odd size: pad 2


def test_bilinear(emu):
    S.bilinear_case(emu)


def test_fusion_block(emu):
    S.fusion_block_case(emu)


def test_loss_and_outlier(emu):
    S.loss_case(emu)


def test_prestep(emu):
    S.prestep_case(emu)


def test_prestep_scale_and_shift(emu):
    S.prestep_st_case(emu)


def test_metrics(emu):
    S.metrics_case(emu)
