"""Per-call latency on a stream of FDDB-sized images (every size new), single caller thread."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jda_amd import synth, api
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, calib); m.save(mp, 8)
rng = np.random.default_rng(5)
sizes = [(int(rng.integers(200, 451)), int(rng.integers(200, 451))) for _ in range(400)]
imgs = [synth.make_frames(1, w, h, seed=3)[0] for w, h in sizes]
c = api.Cascador(mp)
c.detect(imgs[0])
for rep in range(2):
    t0 = time.perf_counter()
    for im in imgs: c.detect(im)
    print("pass %d over 400 sizes (plan cache %s): %.3f ms/call" % (rep, os.environ.get("JDA_PLAN_CACHE", "64"), (time.perf_counter() - t0) / len(imgs) * 1e3))
