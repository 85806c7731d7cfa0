"""k_finish(survivors) of a rank's shard of the FDDB-shaped job: where a window's clocks go (timing build:
   python -m jda_amd.build --timing; JDA_LIB_PATH=jda_amd/libjda_timing.so python tools/shard_finish_timing.py).
   Stamps of a window: start, shape + pixels loaded, [stage 0: after the walk round of carts 512.., after the re-walk of
   carts 0..511 + barrier, after the regression], [stages 1..: after walks + replay, after the regression], end."""
import ctypes as C, os, sys
sys.path.insert(0, ".")
exec(open("./tools/shard_job.py").read().split("for _ in range(3):")[0])
for _ in range(3): job()
buf = np.zeros((65536, 32), np.uint64)
assert api.lib.jdaDebugScanTiming(C.c_void_p(c.h), buf.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
n = (buf[:, 0] & 0xffffffff).astype(int); tag = (buf[:, 0] >> 32).astype(np.int64)
sel = tag == 0x7777
print("k_finish workgroups with stamps:", int(sel.sum()))
idx = np.where(sel)[0]
t0 = buf[idx, 1].astype(np.int64).min()
en = buf[idx, n[idx]].astype(np.int64)
print("launch: first start -> last end %d ticks; starts within %d ticks" % (int(en.max() - t0), int(buf[idx, 1].astype(np.int64).max() - t0)))
for k in sorted(set(n[sel].tolist())):
    s2 = sel & (n == k)
    stp = buf[s2, 1:1 + k].astype(np.int64)
    dt = np.diff(stp, axis=1)
    print("stamps %2d: %5d windows; median ticks per segment %s; total median %d p90 %d; start offset median %d max %d; end max %d" % (
        k, int(s2.sum()), np.median(dt, axis=0).astype(int).tolist(), int(np.median(stp[:, -1] - stp[:, 0])), int(np.percentile(stp[:, -1] - stp[:, 0], 90)),
        int(np.median(stp[:, 0] - t0)), int((stp[:, 0] - t0).max()), int((stp[:, -1] - t0).max())))
