"""Where a k_scan workgroup spends its time (shader-clock stamps; needs `python -m jda_amd.build --timing`).
   JDA_LIB_PATH=jda_amd/libjda_timing.so python tools/scan_timing.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, synth.make_frames(8, 640, 480, seed=0, first=10_000_000)); m.save(mp, 8)
c = api.Cascador(mp)
d = torch.from_numpy(synth.make_frames(256, 640, 480, seed=0)).cuda()
for _ in range(3): c.detect_batch_device(d, keep_results=False)
buf = np.zeros((65536, 32), np.uint64)
assert api.lib.jdaDebugScanTiming(C.c_void_p(c.h), buf.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
n = (buf[:, 0] & 0xffffffff).astype(int); lvl = (buf[:, 0] >> 32).astype(np.int64)
ok = n >= 3
print("blocks with stamps", ok.sum(), "(last launch to write each slot wins)")
for L in sorted(set(lvl[ok].tolist())):
    sel = ok & (lvl == L)
    ns = n[sel]; st = buf[sel, 1:16].astype(np.int64); it = buf[sel, 16:31].astype(np.int64)
    k = int(np.median(ns))
    rows = sel.sum()
    same = ns == k
    dt = np.diff(st[same][:, :k], axis=1)
    print("level %d: %d blocks, median stamps %d; median cycles per segment [prologue, phase0, phase1, ...]:" % (L if L < 2**31 else -1, rows, k),
          np.median(dt, axis=0).astype(int).tolist(), "total", int(np.median(st[same][:, k - 1] - st[same][:, 0])),
          "items at phase ends", np.median(it[same][:, 1:k], axis=0).astype(int).tolist())
