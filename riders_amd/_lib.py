"""ctypes binding of libriders_hip.so (the C ABI declared in include/riders_hip.h).

The prototypes are parsed from the header itself, so the binding can never drift from the ABI and
`tests/test_abi.py` can check that the library exports every declared symbol.

There is NO CPU fallback: if the HIP library is missing `load()` raises, and `ops` refuses host tensors.
`_install_for_tests()` exists only so tests/ can point the same Python layer at the fiber-emulator build of
the very same kernel sources (tests/emu) to check index logic on the GPU-less build container.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "riders_hip.h")
LIB_PATH = os.environ.get("RIDERS_HIP_LIB") or os.path.join(HERE, "libriders_hip.so")   # override: A/B runs of two kernel builds (tools/)

RD_F32, RD_BF16, RD_F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_RELU6 = 0, 1, 2, 3


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("dtype", ctypes.c_int32),
        ("N", ctypes.c_int32), ("Hin", ctypes.c_int32), ("Win", ctypes.c_int32),
        ("C1", ctypes.c_int32), ("C2", ctypes.c_int32),
        ("upsample", ctypes.c_int32), ("H1", ctypes.c_int32), ("W1", ctypes.c_int32),
        ("Cout", ctypes.c_int32), ("KH", ctypes.c_int32), ("KW", ctypes.c_int32),
        ("stride", ctypes.c_int32), ("pad", ctypes.c_int32),
        ("in_dilate", ctypes.c_int32),
        ("OH", ctypes.c_int32), ("OW", ctypes.c_int32),
        ("act", ctypes.c_int32), ("slope", ctypes.c_float),
        ("D1", ctypes.c_int32),
        ("out_reduce2", ctypes.c_int32),
        ("out_d2s", ctypes.c_int32),
        ("in_s2d", ctypes.c_int32),
    ]


class ConvFusion(ctypes.Structure):
    """rd_conv_fusion: consumer-side BatchNorm apply (in_*) and BatchNorm-backward sums in the data-gradient epilogue (bn_*)."""
    _fields_ = [("in_scale", ctypes.c_void_p), ("in_shift", ctypes.c_void_p), ("in_act", ctypes.c_int32), ("in_slope", ctypes.c_float),
                ("bn_y", ctypes.c_void_p), ("bn_scale", ctypes.c_void_p), ("bn_shift", ctypes.c_void_p), ("bn_mean", ctypes.c_void_p),
                ("bn_rstd", ctypes.c_void_p), ("bn_act", ctypes.c_int32), ("bn_slope", ctypes.c_float)]


class PackItem(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("packed", ctypes.c_void_p), ("Cout", ctypes.c_int32), ("Cin", ctypes.c_int32),
                ("KH", ctypes.c_int32), ("KW", ctypes.c_int32), ("mode", ctypes.c_int32), ("dtype", ctypes.c_int32),
                ("Cin_src", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class LwgGemm(ctypes.Structure):
    _fields_ = [("x1", ctypes.c_void_p), ("x2", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("slab", ctypes.c_void_p),
                ("M", ctypes.c_int32), ("C1", ctypes.c_int32), ("C2", ctypes.c_int32), ("Cout", ctypes.c_int32),
                ("nsplit", ctypes.c_int32), ("rows_per_split", ctypes.c_int32)]


class LwgReduce(ctypes.Structure):
    _fields_ = [("slab", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("elems", ctypes.c_int64), ("nsplit", ctypes.c_int32),
                ("accumulate", ctypes.c_int32)]


class WgradReduceItem(ctypes.Structure):
    _fields_ = [("slab", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("Cout", ctypes.c_int32), ("Cin", ctypes.c_int32),
                ("KH", ctypes.c_int32), ("KW", ctypes.c_int32), ("nsplit", ctypes.c_int32), ("accumulate", ctypes.c_int32)]


class DwWgradItem(ctypes.Structure):
    _fields_ = [("partial", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("rows", ctypes.c_int32), ("C", ctypes.c_int32),
                ("KK", ctypes.c_int32), ("accumulate", ctypes.c_int32)]


class ColsumItem(ctypes.Structure):
    _fields_ = [("partial", ctypes.c_void_p), ("out", ctypes.c_void_p), ("rows", ctypes.c_int32), ("C", ctypes.c_int32),
                ("accumulate", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def _ptr_struct(names):
    return [(n, ctypes.c_void_p) for n in names]


class LoftrWeights(ctypes.Structure):
    _fields_ = _ptr_struct(["wq", "wk", "wv", "wm", "w0", "w2", "g1", "b1", "g2", "b2"])


class LoftrSaved(ctypes.Structure):
    _fields_ = _ptr_struct(["q", "k", "v", "att", "mpre", "msg", "hid", "m2pre", "stats"])


class LoftrGrads(ctypes.Structure):
    _fields_ = _ptr_struct(["dout", "dm2pre", "dhid", "dmpre", "datt", "dq", "dk", "dv", "dx", "dsrc", "lnp1", "lnp2", "dg1", "db1",
                            "dg2", "db2"]) + [("accumulate", ctypes.c_int32), ("defer_ln", ctypes.c_int32), ("dsrc_accumulate", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class LnGradItem(ctypes.Structure):
    _fields_ = [("partial", ctypes.c_void_p * 4), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p), ("rows", ctypes.c_int32 * 4),
                ("C", ctypes.c_int32), ("nparts", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_SCALARS = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64,
    "float": ctypes.c_float, "double": ctypes.c_double,
}


def _ctype(decl):
    decl = decl.strip()
    decl = re.sub(r"/\*.*?\*/", "", decl).strip()
    if "*" in decl:
        base = decl.split("*")[0].replace("const", "").strip()
        if base == "rd_conv_desc":
            return ctypes.POINTER(ConvDesc)
        if base == "char":
            return ctypes.c_char_p
        return ctypes.c_void_p
    toks = decl.split()
    ty = toks[0] if len(toks) == 1 else " ".join(toks[:-1])
    ty = ty.replace("const", "").strip()
    return _SCALARS[ty]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"typedef struct.*?\}\s*\w+;", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?m)^\s*((?:const\s+)?[\w]+\s*\*?)\s*(rd_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else _SCALARS[ret.replace("const", "").strip()]
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        protos[name] = (restype, argtypes)
    return protos


_lib = None
ALLOW_HOST_POINTERS = False  # flipped only by _install_for_tests


def _bind(handle):
    for name, (restype, argtypes) in parse_header().items():
        fn = getattr(handle, name)  # AttributeError -> missing symbol, fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    return handle


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "riders_amd: %s not found. Build it with `python -m riders_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        # torch's HIP runtime must be the one that initialises the device: loading this library first (its code objects register with
        # the runtime at dlopen) left the process with "no ROCm-capable device" once torch initialised afterwards
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
        _lib = _bind(ctypes.CDLL(LIB_PATH))
    return _lib


def _install_for_tests(path):
    """TESTS ONLY: bind another build of the same ABI (the tests/emu host emulator) and allow host pointers."""
    global _lib, ALLOW_HOST_POINTERS
    _lib = _bind(ctypes.CDLL(path))
    ALLOW_HOST_POINTERS = True
    return _lib


def _uninstall_for_tests():
    global _lib, ALLOW_HOST_POINTERS
    _lib = None
    ALLOW_HOST_POINTERS = False


def check(rc, what=""):
    if rc != 0:
        msg = load().rd_last_error_string()
        raise RuntimeError("riders_hip %s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))
