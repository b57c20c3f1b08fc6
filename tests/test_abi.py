"""CPU-only checks of the C-ABI boundary: libriders_hip.so (hipcc, gfx950) exists, loads, and exports every symbol
include/riders_hip.h declares; no compute call is made (there is no GPU here)."""
import ctypes
import os

import pytest

from riders_amd import _lib


def test_header_parses_and_lists_entries():
    protos = _lib.parse_header()
    assert len(protos) >= 40
    for must in ("rd_conv_fwd", "rd_conv_wgrad", "rd_linear_attention_fwd", "rd_roi_pool_fwd", "rd_scatter_crops", "rd_adam_step"):
        assert must in protos


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from riders_amd import build
        build.build(verbose=False)
    h = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in _lib.parse_header() if not hasattr(h, n)]
    assert not missing, missing
    h.rd_version.restype = ctypes.c_int
    assert h.rd_version() >= 100


def test_conv_desc_matches_c_layout():
    # 23 x 4-byte fields, no padding; the field names and order are those of the C struct in include/riders_hip.h
    import re
    assert ctypes.sizeof(_lib.ConvDesc) == 92
    src = open(_lib.HEADER).read()
    body = re.search(r"typedef struct rd_conv_desc\s*\{(.*?)\}\s*rd_conv_desc;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
    body = re.sub(r"//[^\n]*", " ", body)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, rest = decl.split(None, 1)
        assert ty in ("int32_t", "float"), decl
        names += [n.strip() for n in rest.split(",")]
    assert names == [f[0] for f in _lib.ConvDesc._fields_], (names, [f[0] for f in _lib.ConvDesc._fields_])


def test_conv_fusion_matches_c_layout():
    """rd_conv_fusion: the ctypes mirror has the header's members in the header's order (pointers 8 bytes, two (int32, float) pairs)."""
    import re
    src = open(_lib.HEADER).read()
    body = re.search(r"typedef struct rd_conv_fusion\s*\{(.*?)\}\s*rd_conv_fusion;", src, flags=re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            names.append(decl.replace("*", " ").split()[-1])
    assert names == [f[0] for f in _lib.ConvFusion._fields_], (names, [f[0] for f in _lib.ConvFusion._fields_])
    assert ctypes.sizeof(_lib.ConvFusion) == 8 * 7 + 4 * 4


def test_png_decode_does_not_touch_the_gpu_runtime():
    """decode_png16 runs in DataLoader workers; it must bind only the host helper library -- neither torch nor libriders_hip.so
    may be loaded by it (checked in a fresh interpreter)."""
    import subprocess
    import sys
    code = ("import sys, numpy as np\n"
            "from riders_amd import data_utils as D\n"
            "z = (np.arange(12 * 20, dtype=np.float32).reshape(12, 20) % 97) / 3.0\n"
            "png = D.encode_png16((z * 256.0).astype(np.uint16))\n"
            "got = D.decode_png16(png)\n"
            "assert got.shape == (12, 20) and got.dtype == np.uint16\n"
            "assert 'torch' not in sys.modules, 'decode_png16 imported torch'\n"
            "maps = open('/proc/self/maps').read()\n"
            "assert 'libriders_hip.so' not in maps and 'libamdhip64' not in maps, 'decode_png16 loaded the HIP library'\n"
            "assert 'libriders_host.so' in maps\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-800:]


def test_product_refuses_host_tensors():
    import torch
    from riders_amd import engine
    _lib._uninstall_for_tests()
    with pytest.raises(RuntimeError):
        engine._stream(torch.zeros(1))
