"""Build recipe for libriders_hip.so (hipcc, gfx950 only).

`python -m riders_amd.build` compiles every translation unit under riders_amd/csrc for gfx950 and links
riders_amd/libriders_hip.so in-tree (the .so travels to the GPU box with the repo snapshot).  hipcc
cross-compiles without a GPU, so this also is the "does it build" check run by __graft_entry__.build().
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libriders_hip.so")
HOST_LIB = os.path.join(HERE, "libriders_host.so")     # host-only helpers (csrc/rd_host.cpp, plain g++): bound without the GPU runtime
HOST_SRC = os.path.join(CSRC, "rd_host.cpp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cpp"))


def units():
    """(source, object, extra flags): every kernel translation unit twice -- 16-bit activations = bf16 (namespace rd) and, with
    -DRD_HALF_F16, = IEEE fp16 (namespace rd_f16, see csrc/rd_common.h); the C ABI layer (rd_api.cpp) once, it dispatches on the dtype code."""
    out = []
    for f in sources():
        src = os.path.join(CSRC, f)
        out.append((src, os.path.join(OBJ, f + ".o"), []))
        if f.endswith(".hip"):
            out.append((src, os.path.join(OBJ, f + ".f16.o"), ["-DRD_HALF_F16"]))
    return out


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


_INC = None


def _local_includes(path, seen):
    """the csrc/*.h files `path` includes, transitively (#include "x.h" only: system and ROCm headers do not change between builds)"""
    import re
    try:
        text = open(path).read()
    except OSError:
        return
    for name in re.findall(r'#\s*include\s+"([^"]+)"', text):
        h = os.path.normpath(os.path.join(os.path.dirname(path), name))
        if os.path.exists(h) and h not in seen:
            seen.add(h)
            _local_includes(h, seen)


def _headers_mtime(src=None):
    """newest header `src` depends on: the headers it includes, transitively (a change in rd_attention_head.h recompiles the two units that
    include it, not all twenty-five; the public header include/riders_hip.h reaches the C ABI layer only)"""
    if src is None:
        hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "riders_hip.h")]
    else:
        seen = set()
        _local_includes(src, seen)
        hs = list(seen)
    return max([os.path.getmtime(h) for h in hs] or [0.0])


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for src, obj, extra in units():
        if force or _newer(src, obj) or _headers_mtime(src) > os.path.getmtime(obj):
            jobs.append((src, obj, extra))

    def cc(job):
        src, obj, extra = job
        cmd = [HIPCC, "-x", "hip"] + FLAGS + extra + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for (src, obj, extra), r in ex.map(cc, jobs):
                if verbose:
                    print("[hipcc] %s%s" % (os.path.basename(src), " (fp16 build)" if extra else ""), flush=True)
                if r.returncode != 0:
                    sys.stderr.write(r.stdout + r.stderr)
                    raise RuntimeError("hipcc failed on %s" % src)
    objs = [obj for _, obj, _ in units()]
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print("[link] %s" % LIB, flush=True)
    build_host(verbose)
    return LIB


def build_host(verbose=False):
    """riders_amd/libriders_host.so: the host-only part of the ABI for processes that must not initialise a GPU (data-loader workers)."""
    if _newer(HOST_SRC, HOST_LIB):
        r = subprocess.run([os.environ.get("CXX", "g++"), "-O2", "-shared", "-fPIC", "-o", HOST_LIB, HOST_SRC], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("host helper build failed")
        if verbose:
            print("[g++] %s" % HOST_LIB, flush=True)
    return HOST_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
