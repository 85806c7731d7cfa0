"""Per-step times of the bench's headline loop (submit/wait) exactly as bench.py times it: a warm-up that drains, then K steps
from an empty pipeline, drained inside the timed region.  Where do the 0.04-0.05 ms per step go that 20 timed steps cost over the
steady state?   python tools/experiments/r06_step_transient.py [depth] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jda_amd import synth, api
import bench
calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
mp = bench.model_path((5, 540, 27, 4), "cascade", 1, calib)
R = 4
ds = [torch.from_numpy(synth.make_frames(256, 640, 480, seed=0, first=i * 256)).cuda() for i in range(R)]
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
c = api.Cascador(mp)
cnt = [0]
def submit():
    cnt[0] += 1
    return c.submit_batch_device(ds[cnt[0] % R])
def loop(n, stamps=None):
    q = [submit() for _ in range(min(depth - 1, n))]
    issued = len(q)
    for i in range(n):
        if issued < n:
            q.append(submit()); issued += 1
        c.wait_batch(q.pop(0), keep_results="packed")
        if stamps is not None: stamps.append(time.perf_counter())
for trial in range(3):
    loop(5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = []
    loop(steps, st)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    d = [(st[0] - t0) * 1e3] + [(st[i] - st[i - 1]) * 1e3 for i in range(1, steps)]
    print("depth %d trial %d: %.4f ms per step over %d; wait-to-wait ms: %s ; after the last wait %.3f" % (depth, trial, (t1 - t0) / steps * 1e3, steps, " ".join("%.2f" % x for x in d), (t1 - st[-1]) * 1e3))
c.close()
