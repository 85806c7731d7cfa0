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
