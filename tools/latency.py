"""Single-frame call latency of the drop-in jdaDetect (plan cached) and with a new frame size per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jda_amd import synth, api
calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, calib); m.save(mp, 8)
c = api.Cascador(mp)
f = synth.make_frames(4, 640, 480, seed=1)
c.detect(f[0])
t0 = time.perf_counter()
for i in range(200): c.detect(f[i % 4])
print("jdaDetect 640x480, plan cached: %.3f ms/call" % ((time.perf_counter() - t0) / 200 * 1e3))
(_, st) = c.detect_batch(f[:1], stats=True)
print("  gpu_ms %.3f scan_ms %.3f host_ms %.3f" % (st["gpu_ms"], st["scan_ms"], st["host_ms"]))
sizes = [(300 + 3 * i, 280 + 2 * i) for i in range(40)]
imgs = [synth.make_frames(1, w, h, seed=2)[0] for w, h in sizes]
t0 = time.perf_counter()
for im in imgs: c.detect(im)
print("jdaDetect new size each call: %.3f ms/call" % ((time.perf_counter() - t0) / len(imgs) * 1e3))
t0 = time.perf_counter()
for im in imgs: c.detect(im)
print("jdaDetect same 40 sizes again (cached): %.3f ms/call" % ((time.perf_counter() - t0) / len(imgs) * 1e3))
