"""SML kernel/host logic under the fiber emulator (see tests/test_emu_ops.py for what these are and are not)."""
from tests import parity_cases_sml as S


def test_effnet_blocks(emu):
    S.effnet_block_case(emu, "ir", 24, 32, 3, 2)
    S.effnet_block_case(emu, "ir", 16, 16, 5, 1, H=7, W=9)
    S.effnet_block_case(emu, "ds", 32, 24, 3, 1, H=8, W=8)


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
