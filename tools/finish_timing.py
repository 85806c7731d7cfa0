"""Where a window of k_finish's second launch (stages 1..T-1) spends its time (shader-clock stamps of the timing build:
   `python -m jda_amd.build --timing`, then JDA_LIB_PATH=jda_amd/libjda_timing.so python tools/finish_timing.py).
   Stamps: start, shape loaded, [first stage: after each walk round, after the barrier, after the regression],
   [later stages: after the walks, after the regression], end."""
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
st = c.last_stats if hasattr(c, "last_stats") else None
buf = np.zeros((65536, 32), np.uint64)
assert api.lib.jdaDebugScanTiming(C.c_void_p(c.h), buf.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
n = (buf[:, 0] & 0xffffffff).astype(int); tag = (buf[:, 0] >> 32).astype(np.int64)
sel = tag == 0x7777
print("k_finish pass-2 workgroups with stamps:", int(sel.sum()))
t0 = buf[sel, 1].astype(np.int64).min()
win = buf[:, 16].astype(int)
tot = (buf[np.arange(65536), np.maximum(n, 1)] - buf[:, 1]).astype(np.int64)
xcd = np.arange(65536) % 8
for wv in sorted(set(win[sel].tolist())):
    s3 = sel & (win == wv)
    print("win %3d: %5d windows, total ticks median %7d p90 %7d max %7d" % (wv, int(s3.sum()), int(np.median(tot[s3])), int(np.percentile(tot[s3], 90)), int(tot[s3].max())))
hw = buf[:, 17]
xcc = ((hw >> 32) & 0xf).astype(int); cu = ((hw >> 8) & 0xf).astype(int); sh = ((hw >> 12) & 1).astype(int); se = ((hw >> 13) & 0x7).astype(int)
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
idx = np.where(sel)[0]
st0 = buf[idx, 1].astype(np.int64); en = buf[idx, n[idx]].astype(np.int64)
print("distinct CUs seen", len(set(cuid[idx].tolist())), "xccs", sorted(set(xcc[idx].tolist())))
for x in sorted(set(xcc[idx].tolist())):
    m = xcc[idx] == x
    print("xcc %d: %5d windows on %2d CUs, first start -> last end %8d ticks; starts within %8d; median window %7d" % (
        x, int(m.sum()), len(set(cuid[idx][m].tolist())), int(en[m].max() - st0[m].min()), int(st0[m].max() - st0[m].min()), int(np.median(en[m] - st0[m]))))
# per CU: windows, span, peak concurrency
rows = []
for c_ in sorted(set(cuid[idx].tolist())):
    m = cuid[idx] == c_
    ev = sorted([(t, 1) for t in st0[m]] + [(t, -1) for t in en[m]])
    cur = pk = 0
    for _, d_ in ev:
        cur += d_; pk = max(pk, cur)
    rows.append((int(m.sum()), int(en[m].max() - st0[m].min()), pk))
rows = np.array(rows)
print("per CU: windows min/median/max %d/%d/%d; span ticks min/median/max %d/%d/%d; peak concurrent windows min/median/max %d/%d/%d" % (
    rows[:, 0].min(), np.median(rows[:, 0]), rows[:, 0].max(), rows[:, 1].min(), np.median(rows[:, 1]), rows[:, 1].max(),
    rows[:, 2].min(), np.median(rows[:, 2]), rows[:, 2].max()))
for k in sorted(set(n[sel].tolist())):
    s2 = sel & (n == k)
    stp = buf[s2, 1:1 + k].astype(np.int64)
    dt = np.diff(stp, axis=1)
    print("stamps %2d: %5d workgroups; median ticks per segment %s; total median %d; start offset median %d max %d; end max %d" % (
        k, int(s2.sum()), np.median(dt, axis=0).astype(int).tolist(), int(np.median(stp[:, -1] - stp[:, 0])),
        int(np.median(stp[:, 0] - t0)), int((stp[:, 0] - t0).max()), int((stp[:, -1] - t0).max())))
