"""Experiment: two host threads with a cascador each, alternating batches of the same workload, vs one caller."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, calib); m.save(mp, 8)
d = torch.from_numpy(synth.make_frames(256, 640, 480, seed=0)).cuda()
K = 60
def run(nthreads):
    cs = [api.Cascador(mp) for _ in range(nthreads)]
    for c in cs:
        for _ in range(3): c.detect_batch_device(d, keep_results="packed")
    torch.cuda.synchronize()
    def work(c, n):
        for _ in range(n): c.detect_batch_device(d, keep_results="packed")
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(c, K // nthreads)) for c in cs]
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    print("%d caller(s): %.3f ms per batch, %.3e windows/s" % (nthreads, el / K * 1e3, 9790720 * K / el))
for n in (3,) * 12 + (4,) * 6:
    run(n)
