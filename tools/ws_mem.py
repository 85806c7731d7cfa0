"""Device memory a cascador takes for configs[2] (256 x 1080p, scale 1.5) through three tickets, and the time of its first
submits: JDA_WS_BOUND=1 (queues sized from the plan's fractions, r06) against 0 (every queue holds every window)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
f2 = synth.make_frames(32, 1920, 1080, seed=0)
mp2 = os.path.join(synth.cache_dir(), "config2_5_540_27_4.model")
if not os.path.exists(mp2):
    m2 = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m2, f2[:4], scale=1.5); m2.save(mp2 + ".tmp", 8); os.replace(mp2 + ".tmp", mp2)
b2 = torch.from_numpy(f2).cuda()
d2 = torch.cat([b2] + [torch.roll(b2, shifts=(131 * j, 257 * j), dims=(1, 2)) for j in range(1, 8)]).contiguous()
del b2
torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info()
c2 = api.Cascador(mp2)
times = []
q = []
t_all = time.perf_counter()
for i in range(8):
    t0 = time.perf_counter()
    if len(q) == 3:
        c2.wait_batch(q.pop(0), keep_results=False)
    q.append(c2.submit_batch_device(d2, 1.5))
    times.append((time.perf_counter() - t0) * 1e3)
while q:
    _, st = c2.wait_batch(q.pop(0), keep_results=False, stats=True)
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info()
t0 = time.perf_counter()
for i in range(6):
    if len(q) == 3:
        c2.wait_batch(q.pop(0), keep_results=False)
    q.append(c2.submit_batch_device(d2, 1.5))
while q:
    c2.wait_batch(q.pop(0), keep_results=False)
steady = (time.perf_counter() - t0) / 6 * 1e3
print("ws_bound=%s: device memory taken by the cascador (3 ticket lanes) %.2f GB; first 8 submit+wait steps (ms) %s; steady %.2f ms/batch; ws_regrows of the last wait %d"
      % (os.environ.get("JDA_WS_BOUND", "1"), (free0 - free1) / 2**30, ["%.1f" % t for t in times], steady, st["ws_regrows"]))
