"""Builds tests/guard/libguard_alloc.so (host code only; test infrastructure, see guard_alloc.cpp)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "guard_alloc.cpp")
OUT = os.path.join(HERE, "libguard_alloc.so")


def build(force=False):
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", "-std=c++17", "-o", OUT, SRC])
    return OUT


if __name__ == "__main__":
    print(build())
