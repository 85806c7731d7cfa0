#!/usr/bin/env python3
"""Runs the PMC calibration kernels (tools/pmc_calib.hip); meant to be wrapped in
rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE)."""
import ctypes, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libpmc_calib.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "pmc_calib.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(here, "pmc_calib.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.pmc_calib_run.argtypes = [ctypes.c_size_t]
if len(sys.argv) > 1 and sys.argv[1] == "gather":
    # the regression gather of BASELINE.json configs[4]: 544-byte rows; table = W of that model (243.7 MB, fits the
    # 256 MB Infinity Cache) and a 1 GiB table (does not); 30,000 waves x 2,000 rows = 32.64 GB of useful bytes each
    lib.pmc_calib_gather.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_uint, ctypes.c_int]
    for tab in (7 * 64000 * 136 * 4, 1 << 30):
        assert lib.pmc_calib_gather(tab, 136, 30000, 2000) == 0
        print("gather table bytes", tab, "useful bytes per dispatch", 30000 * 2000 * 136 * 4)
else:
    nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
    assert lib.pmc_calib_run(nbytes) == 0
    print("calib bytes", nbytes)
