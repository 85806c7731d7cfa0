"""Builds jda_amd/libjda.so (the C-ABI shared library) with hipcc for gfx950.

The library is kept IN-TREE next to this file so it travels to the GPU box
with the repo snapshot.  No torch, no pybind: plain `hipcc -shared`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libjda.so")
SOURCES = ["kernels.hip", "detect.cpp", "model.cpp", "plan.cpp", "post.cpp"]
HEADERS = ["kernels.h", "model.h", "plan.h", "post.h", os.path.join("..", "..", "include", "jda.h")]

# -ffp-contract=off: the cascade must round like the reference's scalar code
# (an FMA changes tree paths); denormals are kept; division is IEEE.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-gpu-flush-denormals-to-zero", "-fvisibility=hidden", "-DJDA_EXPORTS",
         "-Wall", "-Wno-unused-function", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_timing():
    """Investigation build with shader-clock stamps in k_scan (tools/scan_timing.py); never the product."""
    out = os.path.join(HERE, "libjda_timing.so")
    subprocess.check_call([hipcc()] + FLAGS + ["-DJDA_SCAN_TIMING", "-o", out] + [os.path.join(CSRC, s) for s in SOURCES])
    return out


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cmd = [hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--timing" in sys.argv:
        print(build_timing()); sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
