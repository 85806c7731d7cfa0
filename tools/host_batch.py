"""PCIe-inclusive throughput: jdaDetectBatch on 256 host frames (pageable numpy memory) vs frames resident in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, calib); m.save(mp, 8)
c = api.Cascador(mp)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f = synth.make_frames(n, 640, 480, seed=0)
d = torch.from_numpy(f).cuda()
for _ in range(2): c.detect_batch(f)
for name, fn in (("host frames (jdaDetectBatch)", lambda: c.detect_batch(f, stats=True)),
                 ("device frames (jdaDetectBatchDevice)", lambda: c.detect_batch_device(d, stats=True))):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); _, st = fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%s: %d frames, best %.2f ms median %.2f ms per call (gpu_ms %.2f) -> %.0f images/s" % (
        name, n, min(ts), sorted(ts)[2], st["gpu_ms"], n / (sorted(ts)[2] * 1e-3)))
t0 = time.perf_counter(); d2 = torch.from_numpy(f).cuda(); torch.cuda.synchronize()
print("torch pageable H2D of the batch: %.2f ms (%.1f GB/s)" % ((time.perf_counter() - t0) * 1e3, f.nbytes / (time.perf_counter() - t0) / 1e9))
