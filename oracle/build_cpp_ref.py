#!/usr/bin/env python3
"""Opportunistic build of the reference's dialect-CPP detect path (src/jda) -- TEST INFRASTRUCTURE.

Dialect CPP (fp64, round(), cv::Mat patches: reference src/jda/cascador.cpp:166-211,310-477,
cart.cpp:392-404, data.cpp:18-126, btcart.cpp:407-424) is the parity source BASELINE.json's
north_star names, but it cannot be compiled in this image: cascador.cpp:3 and data.cpp:6-7 include
OpenCV, common.cpp:8 includes jsmnpp (an empty, un-pinned submodule), btcart.cpp:3 liblinear.
So dialect CPP's parity is UNPINNED: the HIP kernels are bit-exact against oracle/jda_oracle.c's
restatement only.

This script pins it the day a box has the real dependencies: it looks for REAL OpenCV headers and
libraries (pkg-config opencv4/opencv, or OPENCV_DIR) and a jsmnpp checkout (JSMNPP_DIR, or the
reference's own 3rdparty/jsmnpp if the submodule is populated) and, only if all of them are found,
compiles the reference's own translation units, from where they lie under /root/reference, together
with oracle/cpp_ref_harness.cpp into oracle/_ref/libjda_cppref.so.  It never writes stand-in headers
or stubs: a stand-in build would pin nothing.  tests/test_cpp_reference.py runs when the library
exists and is skipped, with this reason, when it does not.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("JDA_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(HERE, "_ref", "libjda_cppref.so")
UNITS = ["src/jda/cascador.cpp", "src/jda/cart.cpp", "src/jda/btcart.cpp", "src/jda/data.cpp", "src/jda/common.cpp"]


def find_opencv():
    """(cflags, libs) of a real OpenCV, or None."""
    for pkg in ("opencv4", "opencv"):
        if shutil.which("pkg-config"):
            r = subprocess.run(["pkg-config", "--cflags", "--libs", pkg], capture_output=True, text=True)
            if r.returncode == 0:
                parts = r.stdout.split()
                return [p for p in parts if p.startswith("-I")], [p for p in parts if not p.startswith("-I")]
    d = os.environ.get("OPENCV_DIR")
    if d and os.path.exists(os.path.join(d, "include", "opencv2", "core", "core.hpp")):
        return ["-I" + os.path.join(d, "include")], ["-L" + os.path.join(d, "lib"), "-lopencv_core", "-lopencv_imgproc",
                                                     "-lopencv_highgui"]
    for inc in ("/usr/include/opencv4", "/usr/include", "/usr/local/include"):
        if os.path.exists(os.path.join(inc, "opencv2", "core", "core.hpp")):
            return ["-I" + inc], ["-lopencv_core", "-lopencv_imgproc", "-lopencv_highgui"]
    return None


def find_header(name, env, candidates):
    d = os.environ.get(env)
    for c in ([d] if d else []) + candidates:
        if c and os.path.exists(os.path.join(c, name)):
            return c
    return None


def missing():
    """Human-readable list of what keeps the reference's C++ path from compiling here."""
    out = []
    if not os.path.isdir(os.path.join(REF_ROOT, "src", "jda")):
        out.append("reference sources (%s/src/jda)" % REF_ROOT)
    if find_opencv() is None:
        out.append("OpenCV headers/libraries (cascador.cpp:3, data.cpp:6-7)")
    if find_header("jsmn.hpp", "JSMNPP_DIR", [os.path.join(REF_ROOT, "3rdparty", "jsmnpp")]) is None:
        out.append("jsmnpp (common.cpp:8; the reference's 3rdparty/jsmnpp submodule is empty)")
    if find_header(os.path.join("liblinear", "linear.h"), "LIBLINEAR_PARENT", [os.path.join(REF_ROOT, "3rdparty")]) is None:
        out.append("liblinear (btcart.cpp:3; 3rdparty/liblinear is empty)")
    if not os.path.exists(os.path.join(HERE, "cpp_ref_harness.cpp")):
        out.append("oracle/cpp_ref_harness.cpp (the extern \"C\" shim around JoinCascador::Validate / Detect: write it when "
                   "the dependencies above exist, against the real headers)")
    return out


def build():
    m = missing()
    if m:
        raise RuntimeError("dialect CPP reference not buildable here, missing: " + "; ".join(m))
    cflags, libs = find_opencv()
    jsmn = find_header("jsmn.hpp", "JSMNPP_DIR", [os.path.join(REF_ROOT, "3rdparty", "jsmnpp")])
    lin = find_header(os.path.join("liblinear", "linear.h"), "LIBLINEAR_PARENT", [os.path.join(REF_ROOT, "3rdparty")])
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = (["g++", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-I", os.path.join(REF_ROOT, "include"), "-I", jsmn, "-I", lin]
           + cflags + [os.path.join(REF_ROOT, u) for u in UNITS] + [os.path.join(HERE, "cpp_ref_harness.cpp"), "-o", OUT] + libs)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    m = missing()
    if m:
        print("dialect CPP parity stays UNPINNED; missing here:")
        for x in m:
            print("  -", x)
        sys.exit(0 if "--check" in sys.argv else 1)
    print(build())
