"""Throughput of the dialect-CPP entry points on the shipped dimensions (cascade regime):
  uniform batch 640x480: jdaDetectBatchCpp (host frames, H2D inside), jdaDetectBatchCppDevice (resident), method 0
  FDDB-shaped job (2,845 images, the sizes of bench.py's fddb leg): jdaDetectBatchCppRagged[Device], per-image loop
usage: python tools/cpp_bench.py [batch] [--no-pyramid] [--no-loop]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from jda_amd import synth, api
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 256
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, synth.make_frames(8, 640, 480, seed=0, first=10_000_000)); m.save(mp, 8)
c = api.Cascador(mp)
f = synth.make_frames(B, 640, 480, seed=0)
d = torch.from_numpy(f).cuda()


def timed(name, fn, unit_n, reps=3):
    fn(); fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        out, st = fn()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / reps
    ndet = sum(out) if out and isinstance(out[0], int) else sum(len(o["scores"]) for o in out)
    print("dialect CPP %s: %.2f ms, %.3e windows/s, %.0f images/s, avg carts %.1f, gpu_ms %.2f, host_ms %.2f, handoff %d, faces %d, kept %d"
          % (name, el * 1e3, st["patch_n"] / el, unit_n / el, st["average_cart_n"], st["gpu_ms"], st["host_ms"], st["handoff_n"],
             st["face_patch_n"], ndet), flush=True)
    return el


timed("method 1 uniform %d x 640x480, host frames" % B, lambda: c.detect_batch_cpp(f, 20, 5, 1.2, 0.3, True, stats=True), B)
timed("method 1 uniform %d x 640x480, resident" % B, lambda: c.detect_batch_cpp_device(d, 20, 5, 1.2, 0.3, True, stats=True, keep_results=False), B)
if "--no-pyramid" not in sys.argv:
    timed("method 0 (48x48 window, step 5, x1.2), host frames", lambda: c.detect_batch_cpp_pyramid(f, 48, 5, 1.2, 0.3, True, stats=True), B)

# the FDDB-shaped job of bench.py
n_img = 2845
rng = np.random.default_rng(0)
sizes = []
for _ in range(n_img):
    long_side = int(rng.integers(300, 451)); short = int(rng.integers(225, long_side + 1))
    sizes.append((long_side, short) if rng.random() < 0.5 else (short, long_side))
base = synth.make_frames(64, 450, 450, seed=7)
imgs = [np.ascontiguousarray(base[i % 64][:sizes[i][1], :sizes[i][0]]) for i in range(n_img)]
offs, tot = [], 0
for im in imgs:
    offs.append(tot); tot += im.size
buf = np.concatenate([im.reshape(-1) for im in imgs])
ws, hs = [s[0] for s in sizes], [s[1] for s in sizes]
d_buf = torch.from_numpy(buf).cuda()
timed("ragged FDDB-shaped job, resident", lambda: c.detect_ragged_cpp_packed(d_buf, offs, ws, hs, stats=True, keep_results=False), n_img, reps=5)
timed("ragged FDDB-shaped job, packed host buffer", lambda: c.detect_ragged_cpp_packed(buf, offs, ws, hs, stats=True, keep_results=False), n_img, reps=5)
if "--no-loop" not in sys.argv:
    k = 200
    t0 = time.perf_counter()
    for i in range(k):
        c.detect_batch_cpp(imgs[i][None], 20, 5, 1.2, 0.3, True)
    el = time.perf_counter() - t0
    print("dialect CPP per-image loop (the reference's fddb() structure): %.0f images/s over %d images" % (k / el, k))
