"""Negative control of the bounds-check build (python -m jda_amd.build --bounds): with the frames' range cut short by
JDA_BOUNDS_TEST_SHRINK bytes the checker must report violations; with the true range it must report none.
   JDA_LIB_PATH=jda_amd/libjda_bounds.so [JDA_BOUNDS_TEST_SHRINK=20000] python tools/bounds_selftest.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import api, synth
assert hasattr(api.lib, "jdaDebugBoundsReport"), "not the bounds-check build"
p = os.path.join(synth.cache_dir(), "bounds_selftest.model")
synth.make_model(3, 20, 5, 4, seed=3, cart_th=-1.0, norm_every=5).save(p, 8)
c = api.Cascador(p)
d = torch.from_numpy(synth.make_frames(4, 320, 240, seed=1)).cuda()
c.detect_batch_device(d); c.detect_batch_cpp_device(d)
out = (C.c_ulonglong * 16)()
api.lib.jdaDebugBoundsReport.restype = C.c_longlong
n = api.lib.jdaDebugBoundsReport(out)
units = ["k_scan", "k_scan_d", "k_scan_r", "k_scan_dr", "k_scan_p", "k_finish", "k_wide", "k_stage"]
print("shrink %s bytes: %d violation(s) %s" % (os.environ.get("JDA_BOUNDS_TEST_SHRINK", "0"), n,
      {units[i]: (int(out[2 * i + 1]), "site %d line %d" % (out[2 * i] >> 32, out[2 * i] & 0xffffffff)) for i in range(8) if out[2 * i + 1]}))
