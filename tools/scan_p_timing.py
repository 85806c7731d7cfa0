"""Where the waves of a k_scan_p workgroup spend their clocks, per task kind (timing build:
   python -m jda_amd.build --timing; JDA_LIB_PATH=jda_amd/libjda_timing.so JDA_SCAN_P=1 python tools/scan_p_timing.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
mp = bench.model_path((5, 540, 27, 4), "cascade", 1, calib)
c = api.Cascador(mp)
d = torch.from_numpy(synth.make_frames(256, 640, 480, seed=0)).cuda()
for _ in range(3): c.detect_batch_device(d, keep_results=False)
buf = np.zeros((65536, 32), np.uint64)
assert api.lib.jdaDebugScanTiming(C.c_void_p(c.h), buf.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
tag = (buf[:, 0] & 0xffffffff).astype(np.int64); lvl = (buf[:, 0] >> 32).astype(np.int64)
sel = tag == 0x5000
print("persistent workgroups with stamps:", int(sel.sum()), "(the last launch to write a slot wins: the last LDS-tiled level)")
names = ["fresh", "bucket lane=window", "bucket pair", "tile load", "idle", "pick won", "pick lost", "(snapshot read)", "(push to ring)", "(unused)"]
for L in sorted(set(lvl[sel].tolist())):
    s = sel & (lvl == L)
    tot = buf[s, 1].astype(np.float64)
    t = buf[s, 2:12].astype(np.float64); n = buf[s, 12:22].astype(np.float64)
    print("level %d: %d workgroups, %.0f tiles each, median workgroup clocks %.0f" % (L, s.sum(), np.median(buf[s, 22].astype(np.float64)), np.median(tot)))
    wsum = t[:, :7].sum(1)
    for i, nm in enumerate(names):
        print("   %-20s %5.1f %% of wave clocks   %8.0f tasks per workgroup   %8.0f clocks per task" % (
            nm, 100 * np.median(t[:, i] / wsum), np.median(n[:, i]), np.median(t[:, i] / np.maximum(n[:, i], 1))))
