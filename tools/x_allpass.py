"""BASELINE.json configs[4] in the all-pass regime -- the one regime of this path whose weight-row traffic does not
fit the caches: T=7, K=2000, 68 landmarks, depth 6 (W = 243.7 MB, 34.8 MB per stage), every window of a 1080p frame
walks all 14,000 carts.  (k_stage's dense mode does not fit this model: 136 shape coordinates x 256 windows of LDS; the
windows go through k_finish, one wave each, which gathers K 544-byte weight rows per window and stage from L2 / Infinity
Cache / HBM.)
   python tools/x_allpass.py [--frames 1] [--steps 2]
Prints time per frame, carts/s and SURVEY 8(d)'s algorithmic bytes per second.  Wrap in `rocprofv3 --kernel-trace --pmc FETCH_SIZE` for the measured HBM-side bytes."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=1); ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--stages", type=int, default=7, help="T: 7 = configs[4] (W 243.7 MB, inside the 256 MB Infinity Cache); 14 = 487 MB, beyond it")
ap.add_argument("--size", default="1920x1080")
a = ap.parse_args()
dims = (a.stages, 2000, 68, 6)
FW, FH = (int(v) for v in a.size.split("x"))
p = os.path.join(synth.cache_dir(), "x_allpass.model" if a.stages == 7 else "x_allpass_T%d.model" % a.stages)
if not os.path.exists(p):
    synth.make_model(*dims, seed=2).save(p, 4)          # cart_th = NEG_BIG: nothing is rejected
c = api.Cascador(p, "float")
frames = synth.make_frames(a.frames, FW, FH, seed=4)
d = torch.from_numpy(frames).cuda()
kw = dict(th=float("inf"))
c.detect_batch_device(d, keep_results=False, **kw)      # first pass: sparse, switches the plan to dense
_, st = c.detect_batch_device(d, keep_results=False, stats=True, **kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    _, st = c.detect_batch_device(d, keep_results=False, stats=True, **kw)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / a.steps
T, K, L, D = dims
win = st["patch_n"]
carts = st["cart_total_n"]
alg = carts * ((D - 1) * 34 + 16) + win * T * K * 2 * L * 4 + win * 2 * L * 4
rows = win * T * K * 2 * L * 4                 # weight rows the regression gathers: K rows of 2L floats per window and stage
print("X dims all-pass T=%d (W %.1f MB), %d frame(s) %dx%d: %.1f ms per step (gpu %.1f), dense passes %d, %d windows, %.0f carts per window"
      % (T, T * K * 32 * 2 * L * 4 / 1e6, a.frames, FW, FH, el * 1e3, st["gpu_ms"], st["dense_passes"], win, carts / win))
print("  %.3e windows/s  %.3e carts/s  algorithmic bytes (SURVEY 8d, 10.2 MB per window) %.2f TB per step = %.2f TB/s = %.0f %% of the 8 TB/s HBM peak; "
      "weight rows alone %.2f TB = %.2f TB/s = %.0f %% of the peak" % (win / el, carts / el, alg / 1e12, alg / el / 1e12, 100 * alg / el / 8e12, rows / 1e12, rows / el / 1e12, 100 * rows / el / 8e12))
