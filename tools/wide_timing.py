"""Shader-clock stamps (100 MHz) of k_finish_wide for single-frame calls (timing build:
   python -m jda_amd.build --timing; JDA_LIB_PATH=jda_amd/libjda_timing.so python tools/wide_timing.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jda_amd import synth, api
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, synth.make_frames(8, 640, 480, seed=0, first=10_000_000)); m.save(mp, 8)
c = api.Cascador(mp)
f = synth.make_frames(4, 640, 480, seed=1)
for i in range(8): c.detect(f[i % 4])
buf = np.zeros((65536, 32), np.uint64)
assert api.lib.jdaDebugScanTiming(C.c_void_p(c.h), buf.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
n = (buf[:, 0] & 0xffffffff).astype(int); tag = (buf[:, 0] >> 32).astype(np.int64)
sel = np.where(tag == 0x8888)[0]
print("workgroups with stamps:", len(sel))
t0 = min(int(buf[i, 1]) for i in sel)
rows = []
for i in sel:
    st = buf[i, 1:1 + n[i]].astype(np.int64)
    rows.append((int(buf[i, 17]), int(buf[i, 16]), int(st[0] - t0), int(st[-1] - st[0]), np.diff(st).tolist()))
rows.sort()
for stages in sorted(set(r[0] for r in rows)):
    rr = [r for r in rows if r[0] == stages]
    print("stages passed %d: %d windows; start offset (x10 ns) median %d max %d; duration median %d max %d" % (
        stages, len(rr), np.median([r[2] for r in rr]), max(r[2] for r in rr), np.median([r[3] for r in rr]), max(r[3] for r in rr)))
    for r in rr[:3]:
        print("   win %3d start %6d dur %6d segments %s" % (r[1], r[2], r[3], r[4]))
print("kernel span (first start -> last end): %d x10 ns" % max(r[2] + r[3] for r in rows))
