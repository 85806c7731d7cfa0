"""Experiment: several host threads with a cascador each, alternating batches of the same workload, vs one caller.
Variants mimic bench.py's pipelined leg step by step to find what makes it unstable there."""
import os, sys, time, threading, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, calib); m.save(mp, 8)
d = torch.from_numpy(synth.make_frames(256, 640, 480, seed=0)).cuda()
K = 60
def run(nthreads, stats=False, setdev=False, tickets=False, device=None, label=""):
    cs = [api.Cascador(mp, device=device) if device is not None else api.Cascador(mp) for _ in range(nthreads)]
    for c in cs:
        for _ in range(3): c.detect_batch_device(d, keep_results="packed")
    torch.cuda.synchronize()
    ticket = itertools.count(); lock = threading.Lock(); out = []
    def work(c, n):
        if setdev: torch.cuda.set_device(0)
        i = 0
        while True:
            if tickets:
                with lock: j = next(ticket)
                if j >= K: return
            else:
                if i >= n: return
                i += 1
            r = c.detect_batch_device(d, 1.25, 40, -1, -0.5, nms=True, stats=stats, keep_results="packed", frame_offset=0)
            if stats:
                with lock: out.append(r[1])
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(c, K // nthreads)) for c in cs]
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    print("%-28s %d caller(s): %.3f ms per batch" % (label, nthreads, el / K * 1e3))
    for c in cs: c.close()
for rep in range(2):
    run(3, label="plain")
    run(3, stats=True, label="stats")
    run(3, setdev=True, label="set_device")
    run(3, tickets=True, label="tickets")
    run(3, device=0, label="device=0")
    run(3, stats=True, setdev=True, tickets=True, device=0, label="all (bench-like)")
