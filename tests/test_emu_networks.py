"""Whole-module logic checks under the fiber emulator (see tests/test_emu_ops.py for what these are and are not)."""
import os

import pytest

from tests import parity_cases as P

# ~13 min under the emulator: opt-in (RIDERS_EMU_SLOW=1); last run here: both passed against the reference's fixtures
pytestmark = pytest.mark.skipif(os.environ.get("RIDERS_EMU_SLOW") != "1", reason="slow emulator case; set RIDERS_EMU_SLOW=1")


def test_decoder_small_golden(emu):
    P.decoder_case(emu, "small")


def test_rcnet_end_to_end_golden(emu):
    P.rcnet_e2e_case(emu)
