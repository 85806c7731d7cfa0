"""Per-step times of the bench's headline loop (submit/wait, three tickets) from a cold cascador: how many steps until the
period is steady?   python tools/experiments/r06_step_transient.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jda_amd import synth, api
import bench
calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
mp = bench.model_path((5, 540, 27, 4), "cascade", 1, calib)
R = 4
ds = [torch.from_numpy(synth.make_frames(256, 640, 480, seed=0, first=i * 256)).cuda() for i in range(R)]
for trial in range(2):
    c = api.Cascador(mp)
    depth, n = 3, 60
    q = [c.submit_batch_device(ds[j % R]) for j in range(depth - 1)]
    issued = len(q)
    ts = []
    torch.cuda.synchronize()
    for i in range(n):
        t0 = time.perf_counter()
        if issued < n + depth:
            q.append(c.submit_batch_device(ds[issued % R])); issued += 1
        c.wait_batch(q.pop(0), keep_results="packed")
        ts.append((time.perf_counter() - t0) * 1e3)
    while q: c.wait_batch(q.pop(0), keep_results=False)
    print("trial %d per-step ms: %s" % (trial, " ".join("%.2f" % t for t in ts)))
    c.close()
