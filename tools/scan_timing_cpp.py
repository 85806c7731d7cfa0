"""k_scan workgroup stamps (timing build) of ONE chunk of the dialect-CPP FDDB-shaped job:
   python -m jda_amd.build --timing; JDA_LIB_PATH=jda_amd/libjda_timing.so python tools/scan_timing_cpp.py [n_images]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 190
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, synth.make_frames(8, 640, 480, seed=0, first=10_000_000)); m.save(mp, 8)
c = api.Cascador(mp)
c.set_option("ragged_chunk_windows_cpp", 40000000)
c.set_option("ragged_split", 1)
rng = np.random.default_rng(0)
sizes = []
for _ in range(2845):
    long_side = int(rng.integers(300, 451)); short = int(rng.integers(225, long_side + 1))
    sizes.append((long_side, short) if rng.random() < 0.5 else (short, long_side))
sizes = sizes[:n_img]
base = synth.make_frames(64, 450, 450, seed=7)
imgs = [np.ascontiguousarray(base[i % 64][:sizes[i][1], :sizes[i][0]]) for i in range(n_img)]
offs, tot = [], 0
for im in imgs:
    offs.append(tot); tot += im.size
d_buf = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).cuda()
ws, hs = [s[0] for s in sizes], [s[1] for s in sizes]
for _ in range(3):
    out, st = c.detect_ragged_cpp_packed(d_buf, offs, ws, hs, stats=True, keep_results=False)
print("windows %d scan_ms %.3f gpu_ms %.3f launches %d" % (st["patch_n"], st["scan_ms"], st["gpu_ms"], st["scan_launches"]))
buf = np.zeros((65536, 32), np.uint64)
assert api.lib.jdaDebugScanTiming(C.c_void_p(c.h), buf.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
n = (buf[:, 0] & 0xffffffff).astype(int); lvl = (buf[:, 0] >> 32).astype(np.int64)
ok = (n >= 3) & (lvl < 64)
print("workgroups with stamps: %d" % int(ok.sum()))
for L in sorted(set(lvl[ok].tolist())):
    sel = ok & (lvl == L)
    ns = n[sel]; st_ = buf[sel, 1:16].astype(np.int64); it = buf[sel, 16:31].astype(np.int64)
    tot_ = np.array([st_[i, ns[i] - 1] - st_[i, 0] for i in range(len(ns))])
    k = int(np.median(ns)); same = ns == k
    dt = np.diff(st_[same][:, :k], axis=1)
    items = np.median(it[same][:, :k], axis=0).astype(int).tolist()
    print("level %2d: %5d blocks, stamps %2d; median cycles/segment %s; items %s; total median %d mean %d" % (
        L, int(sel.sum()), k, np.median(dt, axis=0).astype(int).tolist(), items, int(np.median(tot_)), int(tot_.mean())))
