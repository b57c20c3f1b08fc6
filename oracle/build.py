"""Compile the oracle's C restatements (gcc) into oracle/_build/liboracle.so.  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "liboracle.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith(".c")]
    if os.path.exists(LIB) and all(os.path.getmtime(s) <= os.path.getmtime(LIB) for s in srcs):
        return LIB
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", LIB] + srcs + ["-lm"], check=True)
    return LIB


if __name__ == "__main__":
    print(build())
