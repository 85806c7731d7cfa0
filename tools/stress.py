"""Soak: many calls of mixed kinds and sizes on one cascador; device memory must stay flat (plan cache bounded,
workspace grow-only) and every call must succeed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, calib); m.save(mp, 8)
c = api.Cascador(mp)
rng = np.random.default_rng(11)
big = synth.make_frames(96, 640, 480, seed=5)
dbig = torch.from_numpy(big).cuda()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
t0 = time.time(); n = 0; free0 = None
while time.time() - t0 < secs:
    kind = int(rng.integers(0, 5))
    if kind == 0:
        w, h = int(rng.integers(60, 500)), int(rng.integers(60, 400))
        c.detect(synth.make_frames(1, w, h, seed=n)[0])
    elif kind == 1:
        k = int(rng.integers(1, 96)); c.detect_batch(big[:k])
    elif kind == 2:
        k = int(rng.integers(1, 96)); c.detect_batch_device(dbig[:k], keep_results="packed")
    elif kind == 3:
        k = int(rng.integers(1, 6)); c.trace(big[:k])
    else:
        k = int(rng.integers(1, 24)); c.detect_batch_cpp(big[:k], 20, 5, 1.2)
    n += 1
    if n == 200:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("calls %d in %.1f s; device memory free after 200 calls %.1f MB, at the end %.1f MB (delta %.1f MB)" % (
    n, time.time() - t0, (free0 or 0) / 2**20, free1 / 2**20, ((free0 or free1) - free1) / 2**20))
