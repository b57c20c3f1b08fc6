"""Parity tests proper: HIP kernels on a real MI355X, called through the C ABI (libriders_hip.so), against the
oracle and against golden vectors produced by the reference.  fp32 path: <= 1e-3 relative (north_star); index
outputs bit-exact."""
import pytest

from tests import parity_cases as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", range(len(P.CONV_CASES)))
def test_conv(gpu, case):
    P.conv_case(gpu, P.CONV_CASES[case])


def test_decoder_block(gpu):
    P.decoder_block_case(gpu)
    P.decoder_block_case(gpu, cin=16, cskip=0, cout=16, hs=(5, 4), hv=(10, 8))
    P.decoder_block_case(gpu, cin=256, cskip=128, cout=256, hs=(7, 3), hv=(15, 6), N=6)


def test_resnet_block(gpu):
    P.resnet_block_case(gpu)
    P.resnet_block_case(gpu, cin=16, cout=16, stride=1)
    P.resnet_block_case(gpu, cin=64, cout=128, stride=2)


def test_linear_attention(gpu):
    P.linear_attention_case(gpu)
    P.linear_attention_case(gpu, N=1, L=7, S=30)
    P.linear_attention_case(gpu, N=64, L=21, S=21)


def test_golden_attention(gpu):
    P.golden_attention_case(gpu)


def test_transformer(gpu):
    P.transformer_case(gpu)
    P.transformer_case(gpu, N=5, n_layers=4)


def test_roi_pool(gpu):
    P.roi_pool_case(gpu)


def test_maxpool(gpu):
    P.maxpool_case(gpu)


def test_labels_loss(gpu):
    P.labels_loss_case(gpu)


def test_scatter_crops(gpu):
    P.scatter_crops_case(gpu)


def test_adam(gpu):
    P.adam_case(gpu)


def test_resnet_encoder_golden(gpu):
    P.resnet_encoder_case(gpu)


def test_decoder_golden(gpu):
    P.decoder_case(gpu, "small")
    P.decoder_case(gpu, "zju")


def test_rcnet_end_to_end_golden(gpu):
    P.rcnet_e2e_case(gpu)


def test_native_library_loaded(gpu):
    """The GPU tests must have run on libriders_hip.so (no silent fallback)."""
    import os
    from riders_amd import _lib
    assert not _lib.ALLOW_HOST_POINTERS
    maps = open("/proc/self/maps").read()
    assert os.path.join("riders_amd", "libriders_hip.so") in maps
