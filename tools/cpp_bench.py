"""Throughput of the dialect-CPP entry points (host frames, so the H2D copy is inside):
method 1 (growing window, fddb defaults) and method 0 (true pyramid) on a 640x480 batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jda_amd import synth, api
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, synth.make_frames(8, 640, 480, seed=0, first=10_000_000)); m.save(mp, 8)
c = api.Cascador(mp)
f = synth.make_frames(B, 640, 480, seed=0)
for name, fn in (("method 1 (min 20, step 5, x1.2)", lambda: c.detect_batch_cpp(f, 20, 5, 1.2, 0.3, True, stats=True)),
                 ("method 0 (48x48 window, step 5, x1.2)", lambda: c.detect_batch_cpp_pyramid(f, 48, 5, 1.2, 0.3, True, stats=True))):
    fn()
    t0 = time.perf_counter()
    for _ in range(3): out, st = fn()
    el = (time.perf_counter() - t0) / 3
    print("dialect CPP %s: %d frames, %.2f ms/batch, %.3e windows/s, %.0f images/s, avg carts %.1f, gpu_ms %.2f"
          % (name, B, el * 1e3, st["patch_n"] / el, B / el, st["average_cart_n"], st["gpu_ms"]))
