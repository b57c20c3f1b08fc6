"""TEST INFRASTRUCTURE ONLY: compile the riders_amd/csrc kernel sources for the HOST against the fiber
emulator in tests/emu/include (see hip/hip_runtime.h there) so kernel index logic can be checked on the
GPU-less build container.  The resulting tests/emu/_build/libriders_emu.so exports the same C ABI as
libriders_hip.so but is never imported by the riders_amd package -- only by tests/test_emu_*.py.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(ROOT, "riders_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libriders_emu.so")
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O1", "-fPIC", "-ffp-contract=off", "-I", os.path.join(HERE, "include"),
         "-Wno-unknown-attributes", "-Wno-unused-value", "-Wno-pass-failed"]


def build(verbose=False, extra=()):
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cpp"))
    sys.path.insert(0, ROOT)
    from riders_amd.build import _local_includes      # "which csrc headers does this unit include, transitively"
    emu_rt = os.path.getmtime(os.path.join(HERE, "include", "hip", "hip_runtime.h"))
    jobs, objs = [], []
    for f in srcs:        # every kernel unit twice: bf16 build and fp16 build (-DRD_HALF_F16, namespace rd_f16), as riders_amd/build.py
        src = os.path.join(CSRC, f)
        seen = set()
        _local_includes(src, seen)
        newest = max([os.path.getmtime(src), emu_rt] + [os.path.getmtime(h) for h in seen])
        for suffix, flags in ((".o", []), (".f16.o", ["-DRD_HALF_F16"])) if f.endswith(".hip") else ((".o", []),):
            obj = os.path.join(OUT, f + suffix)
            objs.append(obj)
            if not os.path.exists(obj) or os.path.getmtime(obj) < newest:
                jobs.append((src, obj, flags))

    def cc(job):
        src, obj, flags = job
        return job, subprocess.run([CXX] + FLAGS + list(extra) + flags + ["-c", src, "-o", obj], capture_output=True, text=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=6) as ex:
            for (src, obj, flags), r in ex.map(cc, jobs):
                if verbose:
                    print("[emu-cc] %s" % os.path.basename(src), flush=True)
                if r.returncode != 0:
                    sys.stderr.write(r.stdout + r.stderr)
                    raise RuntimeError("emulator build failed on %s" % src)
    if jobs or not os.path.exists(LIB):
        r = subprocess.run([CXX, "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("emulator link failed")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
