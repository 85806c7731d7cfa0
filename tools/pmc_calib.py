#!/usr/bin/env python3
"""Runs the PMC calibration kernels (tools/pmc_calib.hip); meant to be wrapped in
rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE)."""
import ctypes, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libpmc_calib.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(here, "pmc_calib.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.pmc_calib_run.argtypes = [ctypes.c_size_t]
nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
assert lib.pmc_calib_run(nbytes) == 0
print("calib bytes", nbytes)
