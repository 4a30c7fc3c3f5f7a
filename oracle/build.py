"""Compile the oracle's plain-C restatement (TEST INFRASTRUCTURE ONLY) with gcc.

    python oracle/build.py   ->  oracle/_build/liboracle.so

There is no compiled reference to build: /root/reference is pure Python (0 lines of C/C++), so oracle/_ref is not
used — the real reference is imported only in the build container by tests/golden/make_golden.py.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "philox_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle.so")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    gcc = shutil.which("gcc") or "gcc"
    subprocess.check_call([gcc, "-O2", "-shared", "-fPIC", "-ffp-contract=off", SRC, "-o", LIB, "-lm"])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
