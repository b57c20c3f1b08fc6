import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU emulator case")


@pytest.fixture(scope="session")
def emu_lib_path():
    """TEST INFRA: host build of the kernel sources against the fiber emulator (tests/emu)."""
    from tests.emu import build_emu
    return build_emu.build()


@pytest.fixture()
def emu(emu_lib_path):
    """Point riders_amd's Python layer at the emulator build for the duration of one test (CPU tensors)."""
    import torch
    from riders_amd import _lib, engine
    _lib._install_for_tests(emu_lib_path)
    engine.clear_caches()
    engine._roi_flags.clear()      # the compact RoI arg-max's sticky overflow flag does not travel between tests
    engine.L().rd_clear_options()  # ... nor do routing options a failed test left set
    engine.set_compute_dtype("fp32")
    yield torch.device("cpu")
    engine.L().rd_clear_options()
    engine.clear_caches()
    _lib._uninstall_for_tests()


@pytest.fixture()
def gpu():
    import torch
    from riders_amd import _lib, engine
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a GPU")
    _lib._uninstall_for_tests()
    _lib.load()
    engine.clear_caches()
    engine._roi_flags.clear()
    engine.L().rd_clear_options()
    engine.set_compute_dtype("fp32")
    yield torch.device("cuda:0")
    engine.L().rd_clear_options()
    engine.clear_caches()
