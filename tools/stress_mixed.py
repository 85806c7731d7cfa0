"""Soak of the round-5 paths under concurrency: three threads on ONE cascador -- ragged jobs (persistent scan, RAGGED form),
uniform batches through submit/wait, single frames -- for N seconds; every result is compared with the one computed up
front, device memory must stay flat.   python tools/stress_mixed.py [seconds]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
import bench
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
mp = bench.model_path((5, 540, 27, 4), "cascade", 1, calib)
c = api.Cascador(mp)
rng = np.random.default_rng(5)
base = synth.make_frames(16, 450, 450, seed=7)
imgs = [np.ascontiguousarray(base[i % 16][:int(rng.integers(225, 451)), :int(rng.integers(300, 451))]) for i in range(900)]
batch = torch.from_numpy(synth.make_frames(128, 640, 480, seed=3)).cuda()
frames = synth.make_frames(8, 640, 480, seed=4)
same = lambda a, b: all(np.array_equal(np.ascontiguousarray(a[k]).view(np.uint8), np.ascontiguousarray(b[k]).view(np.uint8)) for k in ("bboxes", "scores", "shapes"))
want_r = c.detect_ragged(imgs)
want_b = c.detect_batch_device(batch)
want_f = [c.detect(f) for f in frames]
errs, counts = [], [0, 0, 0]
stop = [0.0]
def ragged():
    while time.time() < stop[0]:
        got = c.detect_ragged(imgs)
        if not all(same(a, b) for a, b in zip(got, want_r)): errs.append("ragged job differs")
        counts[0] += 1
def tickets():
    q = [c.submit_batch_device(batch)]
    while time.time() < stop[0]:
        q.append(c.submit_batch_device(batch))
        got = c.wait_batch(q.pop(0))
        if not all(same(a, b) for a, b in zip(got, want_b)): errs.append("batch differs")
        counts[1] += 1
    c.wait_batch(q.pop(0))
def singles():
    i = 0
    while time.time() < stop[0]:
        if not same(c.detect(frames[i % 8]), want_f[i % 8]): errs.append("frame differs")
        i += 1; counts[2] += 1
def run(duration):
    stop[0] = time.time() + duration
    ths = [threading.Thread(target=f) for f in (ragged, tickets, singles)]
    for t in ths: t.start()
    for t in ths: t.join()
run(3.0)                                   # (the lanes of the three callers are created here)
torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
counts[:] = [0, 0, 0]
run(secs)
torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]
print("ragged jobs %d, batches %d, single frames %d in %.0f s; errors %d %s; device memory delta after the warm-up %.1f MB; last error '%s'" %
      (counts[0], counts[1], counts[2], secs, len(errs), errs[:3], (free0 - free1) / 2**20, api.last_error()))
sys.exit(1 if errs else 0)
