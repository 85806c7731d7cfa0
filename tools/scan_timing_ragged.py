"""k_scan workgroup stamps (timing build) for the SAME 256 frames through the uniform entry and through the ragged entry:
   JDA_LIB_PATH=jda_amd/libjda_timing.so JDA_LANES=1 python tools/scan_timing_ragged.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, synth.make_frames(8, 640, 480, seed=0, first=10_000_000)); m.save(mp, 8)
c = api.Cascador(mp)
c.set_option("ragged_chunk_windows", 12000000)
frames = synth.make_frames(256, 640, 480, seed=0)
d = torch.from_numpy(frames).cuda()
offs = [i * 640 * 480 for i in range(256)]; ws = [640] * 256; hs = [480] * 256


def dump(tag):
    buf = np.zeros((65536, 32), np.uint64)
    assert api.lib.jdaDebugScanTiming(C.c_void_p(c.h), buf.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
    n = (buf[:, 0] & 0xffffffff).astype(int); lvl = (buf[:, 0] >> 32).astype(np.int64)
    ok = (n >= 3) & (lvl < 64)
    for L in sorted(set(lvl[ok].tolist())):
        sel = ok & (lvl == L)
        ns = n[sel]; st = buf[sel, 1:16].astype(np.int64)
        k = int(np.median(ns)); same = ns == k
        dt = np.diff(st[same][:, :k], axis=1)
        print("%s level %d: %d blocks, median cycles per segment [prologue, phase0, ...]: %s total %d" % (
            tag, L, int(sel.sum()), np.median(dt, axis=0).astype(int).tolist(), int(np.median(st[same][:, k - 1] - st[same][:, 0]))))


for _ in range(2):
    c.detect_batch_device(d, keep_results=False)
dump("uniform")
for _ in range(2):
    c.detect_ragged_packed(d.view(-1), offs, ws, hs, keep_results=False)
dump("ragged ")
