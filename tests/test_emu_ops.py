"""Kernel/host LOGIC checks on the GPU-less build container: the riders_amd Python layer driving a host build of
the same kernel sources under the fiber emulator (tests/emu), compared with the oracle.  These do not count as
parity evidence (that is tests/test_gpu_parity.py on a real MI355X); they keep index math, LDS layouts, the tape
and the module wiring honest between GPU runs."""
import pytest

from tests import parity_cases as P


@pytest.mark.parametrize("case", range(len(P.CONV_CASES)))
def test_conv(emu, case):
    P.conv_case(emu, P.CONV_CASES[case])


@pytest.mark.parametrize("case", range(len(P.PATCH_CONV_CASES)))
def test_patch_conv(emu, case):
    with P.force_patch_conv():
        P.conv_case(emu, P.PATCH_CONV_CASES[case])


def test_patch_conv_decoder_and_bf16(emu):
    with P.force_patch_conv():
        P.decoder_block_case(emu, cin=32, cskip=32, cout=32)
        P.bf16_exact_conv_case(emu, cin=64, cout=64, k=3, s=1, H=9, W=19, N=1)
        P.bf16_exact_conv_case(emu, cin=32, cout=128, k=3, s=1, N=1, up=((4, 3), (17, 6)), cin2=32)
        # narrow-layer kernel, bf16: 2 / 4 / 8 slots per pixel, dual-destination dgrad, upsample + concat gather
        P.bf16_exact_conv_case(emu, cin=16, cout=16, k=3, s=1, H=17, W=9, N=1)
        P.bf16_exact_conv_case(emu, cin=16, cout=16, k=3, s=1, N=1, up=((4, 3), (9, 17)), cin2=16)
        P.bf16_exact_conv_case(emu, cin=64, cout=8, k=3, s=1, H=8, W=5, N=2)
        P.bf16_exact_conv_case(emu, cin=32, cout=64, k=3, s=1, H=12, W=17, N=1)     # 64 output channels: weights in LDS
        P.bf16_exact_conv_case(emu, cin=16, cout=48, k=3, s=1, H=9, W=10, N=2)
    with P.force_patch_conv(g8=1):   # 8 persistent blocks: tile loop, LDS double buffer, register prefetch
        # patch-staged kernel, several tiles per persistent block: cross-tile patch / weight prefetch, per-tile statistics rows
        P.conv_case(emu, dict(cin=32, cout=32, k=3, s=1, H=20, W=40, N=3, bn=True))
        P.bf16_exact_conv_case(emu, cin=128, cout=160, k=3, s=1, H=17, W=33, N=2)
        P.bf16_exact_conv_case(emu, cin=32, cout=16, k=3, s=1, H=33, W=25, N=3)
        P.conv_case(emu, dict(cin=16, cout=16, k=3, s=1, H=30, W=20, N=3, bn=True))


def test_upsample_fused_dgrad(emu):
    P.upsample_fused_dgrad_cases(emu)


def test_grad_add_in_data_gradient_epilogue(emu):
    P.grad_add_cases(emu)


def test_lazy_batchnorm_is_bit_identical(emu):
    P.lazy_bn_cases(emu)


def test_frag_conv(emu):
    P.frag_conv_cases(emu, quick=True)


def test_frag32_conv(emu):
    P.frag32_cases(emu, quick=True)


def test_wgrad_fit(emu):
    P.wgrad_fit_cases(emu, quick=True)


def test_up2_on_source(emu):
    P.up2_cases(emu, quick=True)


def test_bn_head_fused(emu):
    P.bn_head_cases(emu, quick=True)


def test_bn_bwd_sums_in_dgrad_epilogue(emu):
    P.bn_bwd_fused_cases(emu, quick=True)


def test_decoder_block(emu):
    P.decoder_block_case(emu)
    P.decoder_block_case(emu, cin=16, cskip=0, cout=16, hs=(5, 4), hv=(10, 8))


def test_resnet_block(emu):
    P.resnet_block_case(emu)
    P.resnet_block_case(emu, cin=16, cout=16, stride=1)


def test_linear_attention(emu):
    P.linear_attention_case(emu)
    P.linear_attention_case(emu, N=1, L=7, S=30)


def test_golden_attention(emu):
    P.golden_attention_case(emu)


def test_transformer(emu):
    P.transformer_case(emu)


def test_roi_pool_compact_argmax(emu):
    P.roi_pool_compact_case(emu)


def test_roi_pool(emu):
    P.roi_pool_case(emu)


def test_maxpool(emu):
    P.maxpool_case(emu)


def test_labels_loss(emu):
    P.labels_loss_case(emu)


def test_scatter_crops(emu):
    P.scatter_crops_case(emu)


def test_adam(emu):
    P.adam_case(emu)


def test_bf16_exact_conv(emu):
    P.bf16_exact_conv_case(emu)
    P.bf16_exact_conv_case(emu, cin=32, cout=64, k=3, s=2, H=10, W=13)
    P.bf16_exact_conv_case(emu, cin=8, cout=1, k=3, s=1, H=6, W=5)
    P.bf16_exact_conv_case(emu, cin=16, cout=16, k=3, s=1, N=1, up=((4, 3), (9, 6)), cin2=8)
    P.bf16_exact_conv_case(emu, cin=3, cout=32, k=7, s=2, H=14, W=12, N=1)
    P.bf16_exact_conv_case(emu, cin=128, cout=128, k=1, s=1, H=5, W=1, N=2)


def test_bf16_wgrad_transpose_read(emu):
    """bf16 weight gradient of narrow 3x3 layers through the LDS transpose read (rd_wgrad3x3.hip): every (Cin tile, Cout tile, tile
    width) variant, the element-wise dY staging of the 1-channel head, ragged tiles and several tiles per persistent block."""
    P.bf16_exact_conv_case(emu, cin=16, cout=1, k=3, s=1, H=9, W=20, N=1)
    P.bf16_exact_conv_case(emu, cin=16, cout=32, k=3, s=1, H=8, W=33, N=1)
    P.bf16_exact_conv_case(emu, cin=32, cout=24, k=3, s=1, H=11, W=16, N=2)
    P.bf16_exact_conv_case(emu, cin=64, cout=32, k=3, s=1, H=9, W=17, N=1)
    # wide layers: (64-channel input slice) x (32-channel output slice) blocks, concat boundary inside a slice, partial last output slice
    P.bf16_exact_conv_case(emu, cin=128, cout=64, k=3, s=1, H=17, W=16, N=1)
    P.bf16_exact_conv_case(emu, cin=96, cout=40, k=3, s=1, N=1, up=((5, 7), (15, 14)), cin2=32)
    P.bf16_exact_conv_case(emu, cin=192, cout=16, k=3, s=1, H=8, W=14, N=2)
    with P.force_patch_conv(g8=1):
        P.bf16_exact_conv_case(emu, cin=16, cout=16, k=3, s=1, H=40, W=50, N=2)
        P.bf16_exact_conv_case(emu, cin=16, cout=16, k=3, s=1, N=2, up=((13, 9), (27, 64)), cin2=16)


def test_grouped_linear_wgrad(emu):
    """rd_linear_wgrad_batch (deferred, grouped weight gradients of 1x1 / linear layers): bf16 transpose-read and fp32 paths, two-source
    concat, several token splits, ragged last stage."""
    P.bf16_exact_conv_case(emu, cin=64, cout=192, k=1, s=1, H=9, W=7, N=3, cin2=64)
    P.bf16_exact_conv_case(emu, cin=128, cout=64, k=1, s=1, H=41, W=17, N=1)
    P.conv_case(emu, dict(cin=128, cout=64, k=1, s=1, H=13, W=11, N=2, bn=False, act=None))
    P.conv_case(emu, dict(cin=64, cout=128, k=1, s=1, H=30, W=23, N=1, bn=True))
    # ragged edge tiles: channel counts that are multiples of the 16-byte vector only
    P.bf16_exact_conv_case(emu, cin=24, cout=144, k=1, s=1, H=9, W=7, N=2)
    P.bf16_exact_conv_case(emu, cin=64, cout=40, k=1, s=1, H=11, W=5, N=1, cin2=24)
    P.conv_case(emu, dict(cin=36, cout=100, k=1, s=1, H=13, W=11, N=2, bn=True))


def test_pack_batch(emu):
    P.pack_batch_case(emu)


def test_loftr_unfused_path(emu):
    """The per-op LoFTR path (separate projection / attention / LayerNorm launches) stays covered now that eligible layers run fused."""
    from riders_amd import engine
    engine.set_fused_loftr(False)
    try:
        P.golden_attention_case(emu)
        P.transformer_case(emu)
    finally:
        engine.set_fused_loftr(True)


def test_bn_generic_channel_counts(emu):
    """BatchNorm / activation passes for channel counts whose 16-byte vector count does not divide 256 (EfficientNet-Lite3 widths)."""
    P.conv_case(emu, dict(cin=24, cout=40, k=1, s=1, H=13, W=11, N=2, bn=True))
    P.conv_case(emu, dict(cin=16, cout=144, k=1, s=1, H=9, W=10, N=2, bn=True))
    P.conv_case(emu, dict(cin=8, cout=232, k=3, s=1, H=6, W=7, N=1, bn=True))
    P.conv_case(emu, dict(cin=8, cout=288, k=1, s=1, H=9, W=18, N=2, bn=True))     # > 256 channels: short per-block pixel ranges in the backward reduce
    P.conv_case(emu, dict(cin=8, cout=816, k=1, s=1, H=5, W=7, N=1, bn=True))


def test_stem_padded_channels(emu):
    """3-channel stems without an input gradient run zero-padded to one 16-byte vector (rd_pad_channels / rd_conv_pack_weights_padded /
    rd_unpad_weight_grad): forward, BatchNorm and the weight gradient must match the oracle on the original 3-channel weights."""
    P.conv_case(emu, dict(cin=3, cout=32, k=7, s=2, H=20, W=18, N=2, bn=True, no_input_grad=True))
    P.conv_case(emu, dict(cin=3, cout=32, k=3, s=2, H=15, W=14, N=1, bn=True, no_input_grad=True))
    P.conv_case(emu, dict(cin=5, cout=16, k=3, s=1, H=9, W=11, N=2, bn=False, act=None, no_input_grad=True))
    P.stem_kernel_cases(emu)


def test_batch_transforms(emu):
    P.transforms_case(emu)


def test_batch_transforms_noise_and_both_flips(emu):
    P.transforms_all_case(emu)


def test_projection_scatter(emu):
    P.projection_case(emu)


def test_lidar_interpolation(emu):
    P.interpolation_case(emu)
    P.interpolator_case(emu)


def test_fp16_build_exact(emu):
    """fp16 build of the kernels on the emulator (conversion + a few exact convolution families; the full list runs on the GPU)."""
    import torch
    h = torch.float16
    P.bf16_exact_conv_case(emu, half=h)
    P.bf16_exact_conv_case(emu, half=h, cin=16, cout=16, k=3, s=1, N=1, up=((4, 3), (9, 6)), cin2=8)
    P.bf16_exact_conv_case(emu, half=h, cin=64, cout=32, k=3, s=1, H=9, W=17, N=1)
    P.bf16_exact_conv_case(emu, half=h, cin=24, cout=144, k=1, s=1, H=13, W=11, N=2)
    with P.bf16_mode("fp16"):
        P.conv_case(emu, dict(cin=16, cout=16, k=3, s=1, H=12, W=10, N=2, bn=True), tol=4e-3)


def test_inference_driver(emu):
    P.inference_driver_case(emu)


def test_roi_pool_known_answers(emu):
    P.roi_pool_kat_case(emu)


def test_roi_pool_gather_backward(emu):
    P.roi_pool_stress_case(emu)


def test_roi_pool_tile_parity_classes(emu):
    P.roi_pool_tile_deterministic_case(emu)


def test_direct_pointwise_conv(emu):
    """1x1 layers with a short K axis through conv1x1_direct_kernel (pixel fragments straight from memory, weights in registers)."""
    with P.force_direct_1x1():
        P.conv_case(emu, dict(cin=24, cout=144, k=1, s=1, H=13, W=11, N=2, bn=True))             # fp32: two k-steps, two channel blocks
        P.conv_case(emu, dict(cin=16, cout=40, k=1, s=1, H=9, W=15, N=1, bn=False, act=None))    # fp32: one k-step
        P.bf16_exact_conv_case(emu, cin=24, cout=144, k=1, s=1, H=13, W=11, N=2)                  # bf16: one k-step
        P.bf16_exact_conv_case(emu, cin=48, cout=288, k=1, s=1, H=7, W=19, N=1)                   # bf16: two k-steps, three channel blocks
        P.bf16_exact_conv_case(emu, cin=64, cout=8, k=1, s=1, H=10, W=13, N=1)


def test_roi_pool_gather_rcnet_geometry(emu):
    P.roi_pool_gather_rcnet_geometry_case(emu)


def test_wgrad_reduce_batch(emu):
    P.wgrad_reduce_batch_case(emu)


def test_streaming_weight_gradient_of_few_channel_layers(emu):
    P.tiny_wgrad_cases(emu)


def test_point_mlp(emu):
    P.point_mlp_case(emu, R=6)


def test_stride2_dgrad_parity_classes(emu):
    P.stride2_dgrad_cases(emu)


def test_skinny_linear(emu):
    P.skinny_linear_cases(emu)



def test_workspace_bytes_dispatch(emu):
    """rd_workspace_bytes(op, desc): one sizing entry over the per-op helpers; unknown op / bad descriptor -> -1."""
    import ctypes
    from riders_amd import engine
    lib = engine.L()
    d = engine._desc(engine.RD_BF16, 2, 24, 16, 64, 0, False, 24, 16, 128, 3, 3, 1, 1, 1, 24, 16, engine.ACT_NONE, 0.2, 128)
    ref = ctypes.byref(d)
    assert lib.rd_workspace_bytes(0, ref) == lib.rd_conv_wgrad_workspace_bytes(ref) > 0
    assert lib.rd_workspace_bytes(1, ref) == lib.rd_conv_stats_rows(ref) * 128 * 2 * 4
    assert lib.rd_workspace_bytes(2, ref) == lib.rd_conv_packed_elems(128, 9 * 64, engine.RD_BF16) * 2
    assert lib.rd_workspace_bytes(3, ref) == lib.rd_conv_packed_elems(64, 9 * 128, engine.RD_BF16) * 2
    assert lib.rd_workspace_bytes(9, ref) == -1 and b"workspace_bytes" in lib.rd_last_error_string()
    assert lib.rd_workspace_bytes(0, None) == -1


def test_pointwise_gemm_kernel(emu):
    P.pw_gemm_cases(emu, quick=True)


def test_bn_one_launch_wide_layers(emu):
    P.bn_slab_cases(emu)
