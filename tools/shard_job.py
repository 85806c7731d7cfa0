"""One rank's shard of the FDDB-shaped job (356 of the 2,845 images, as shard_range gives rank 0 of 8) through
jdaDetectBatchRaggedDevice, a few times: for rocprofv3 --kernel-trace (tools/shard_timeline.py prints the last job's timeline).
usage: python tools/shard_job.py [reps] [world] [rank]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from jda_amd import synth, api, dist as jdist
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, synth.make_frames(8, 640, 480, seed=0, first=10_000_000)); m.save(mp, 8)
import _dummy_streams; _dummy_streams.make()
c = api.Cascador(mp)
rng = np.random.default_rng(0)
sizes = []
for _ in range(2845):
    long_side = int(rng.integers(300, 451)); short = int(rng.integers(225, long_side + 1))
    sizes.append((long_side, short) if rng.random() < 0.5 else (short, long_side))
weights = [api.count_windows(w_, h_)[0] for (w_, h_) in sizes]
lo, hi = jdist.shard_range_weighted(weights, rank, world) if not os.environ.get("SHARD_BY_COUNT") else jdist.shard_range(2845, rank, world)
base = synth.make_frames(64, 450, 450, seed=7)
imgs = [np.ascontiguousarray(base[i % 64][:sizes[i][1], :sizes[i][0]]) for i in range(lo, hi)]
offs, tot = [], 0
for im in imgs:
    offs.append(tot); tot += im.size
d_buf = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).cuda()
ws, hs = np.array([sizes[i][0] for i in range(lo, hi)], np.int32), np.array([sizes[i][1] for i in range(lo, hi)], np.int32)
offs = np.array(offs, np.uint64)          # (the job description as arrays: the binding passes them on as they are)
job = lambda: c.detect_ragged_packed(d_buf, offs, ws, hs, stats=True, keep_results="packed", frame_offset=lo)
for _ in range(3):
    job()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    rows, st = job()
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / reps
print("shard %d/%d: %d images, %d windows: %.3f ms per job, gpu_ms %.3f scan_ms %.3f host_ms %.3f launches %d handoff %d rows %d"
      % (rank, world, hi - lo, st["patch_n"], el * 1e3, st["gpu_ms"], st["scan_ms"], st["host_ms"], st["scan_launches"], st["handoff_n"], len(rows)))
