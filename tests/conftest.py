import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


def bits(a):
    """View floats as integers so comparisons are bit-exact (NaN-safe, -0 != +0)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a.view(np.uint32)
    if a.dtype == np.float64:
        return a.view(np.uint64)
    return a


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(bits(a), bits(b))


@pytest.fixture(scope="session")
def built():
    """Native pieces present (build them here if this is a fresh checkout)."""
    from jda_amd import build as lib_build
    from oracle import build as oracle_build
    lib_build.build()
    oracle_build.build_oracle()
    return True


@pytest.fixture()
def model_file(tmp_path):
    from jda_amd import synth

    def make(dims, real_bytes=8, **kw):
        m = synth.make_model(*dims, **kw)
        p = str(tmp_path / ("m_%d_%d_%d_%d_%d.model" % (tuple(dims) + (real_bytes,))))
        m.save(p, real_bytes)
        return p, m
    return make


# dimension sets that also have a compiled reference build (oracle/build.py REF_DIMS)
TINY_DIMS = [(2, 8, 5, 3), (3, 20, 5, 4), (2, 6, 4, 6), (1, 4, 3, 2), (3, 70, 9, 5), (2, 64, 68, 6)]
S_DIMS = (5, 540, 27, 4)


# ---- bounds-check build (python -m jda_amd.build --bounds; JDA_LIB_PATH=jda_amd/libjda_bounds.so) ----
# The pool refuses GPU AddressSanitizer; libjda_bounds.so checks every model- / plan-derived device access against the
# extent of what it reads (csrc/kernels_common.h: Bc).  When the suite runs on that build, every test ends with a look at
# the violation words: a test whose kernels touched anything out of bounds fails, naming the site and the source line.
BC_SITES = {1: "scan pixel (LDS tile)", 2: "scan pixel (frame)", 3: "tile load", 4: "finish pixel (frame)", 5: "finish pixel (LDS tile)",
            6: "weight row", 7: "node table", 8: "window tile load", 9: "stage-0 table", 10: "k_stage pixel", 11: "queue"}
BC_UNITS = ["k_scan", "k_scan_d", "k_scan_r", "k_scan_dr", "k_scan_p", "k_finish", "k_wide", "k_stage"]


@pytest.fixture(autouse=True)
def _bounds_report(request):
    yield
    if "jda_amd.api" not in sys.modules:
        return
    api = sys.modules["jda_amd.api"]
    if not hasattr(api.lib, "jdaDebugBoundsReport"):
        return
    import ctypes as C
    out = (C.c_ulonglong * 16)()
    api.lib.jdaDebugBoundsReport.restype = C.c_longlong
    total = api.lib.jdaDebugBoundsReport(out)
    log = os.environ.get("JDA_BOUNDS_LOG")
    if log:
        with open(log, "a") as f:
            f.write("%s %d\n" % (request.node.nodeid, total))
    if total:
        what = ["%s: %d violation(s), first at %s line %d" % (BC_UNITS[i], out[2 * i + 1], BC_SITES.get(out[2 * i] >> 32, "?"), out[2 * i] & 0xffffffff)
                for i in range(8) if out[2 * i + 1]]
        pytest.fail("device bounds check: " + "; ".join(what))
