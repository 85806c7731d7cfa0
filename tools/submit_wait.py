"""Submit/wait pipeline from one thread vs the synchronous call: parity and ms per batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, calib); m.save(mp, 8)
d = torch.from_numpy(synth.make_frames(256, 640, 480, seed=0)).cuda()
c = api.Cascador(mp)
want = c.detect_batch_device(d, keep_results="packed")
for _ in range(3): c.detect_batch_device(d, keep_results="packed")
K = 60
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): c.detect_batch_device(d, keep_results="packed")
    a = (time.perf_counter() - t0) / K * 1e3
    ahead = int(os.environ.get("AHEAD", "1"))
    t0 = time.perf_counter()
    q = [c.submit_batch_device(d) for _ in range(min(ahead, K))]
    issued = len(q)
    got = None
    for i in range(K):
        if issued < K:
            q.append(c.submit_batch_device(d)); issued += 1
        got, st = c.wait_batch(q.pop(0), stats=True, keep_results="packed")
    b = (time.perf_counter() - t0) / K * 1e3
    same = got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    print("sync %.3f ms per batch | submit/wait %.3f ms per batch (parity %s, gpu_ms %.2f call_ms %.2f)" % (a, b, same, st["gpu_ms"], st["call_ms"]))
