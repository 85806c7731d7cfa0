"""The FDDB-shaped job (2,845 images) through jdaDetectBatchCppRaggedDevice, images resident, a few times: for rocprofv3.
usage: python tools/cpp_job.py [reps] [n_images]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from jda_amd import synth, api
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 2845
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
sizes = sizes[:n_img]
base = synth.make_frames(64, 450, 450, seed=7)
imgs = [np.ascontiguousarray(base[i % 64][:sizes[i][1], :sizes[i][0]]) for i in range(n_img)]
offs, tot = [], 0
for im in imgs:
    offs.append(tot); tot += im.size
buf = np.concatenate([im.reshape(-1) for im in imgs])
ws, hs = [s[0] for s in sizes], [s[1] for s in sizes]
d_buf = torch.from_numpy(buf).cuda()
job = lambda: c.detect_ragged_cpp_packed(d_buf, offs, ws, hs, stats=True, keep_results="packed")
job(); job()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    out, st = job()
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / reps
print("CPP ragged job, %d images resident: %.2f ms, %.0f images/s, %.3e windows/s, gpu_ms %.2f scan_ms %.2f host_ms %.2f launches %d handoff %d faces %d"
      % (n_img, el * 1e3, n_img / el, st["patch_n"] / el, st["gpu_ms"], st["scan_ms"], st["host_ms"], st["scan_launches"], st["handoff_n"], st["face_patch_n"]))
print("C call %.2f ms of the %.2f ms per job (the rest: the binding packs %d rows of %d doubles)" % (st["call_ms"], el * 1e3, len(out), out.shape[1] if len(out) else 0))
c.set_option("debug_times", 2)
job()
print("stage_done", st["stage_done_n"][:5], "cart_total", st["cart_total_n"], "scan_cart", st["scan_cart_n"])
