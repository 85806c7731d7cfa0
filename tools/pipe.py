"""The bench's headline leg alone: submit/wait, two batches in flight, four rotating resident batches (for kernel
timelines and counter passes of exactly that stream).   PIPE_STEPS=40 python tools/pipe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jda_amd import synth, api
import bench
calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
mp = bench.model_path((5, 540, 27, 4), "cascade", 1, calib)
R = 4
ds = [torch.from_numpy(synth.make_frames(256, 640, 480, seed=0, first=i * 256)).cuda() for i in range(R)]
c = api.Cascador(mp)
K = int(os.environ.get("PIPE_STEPS", "40"))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = c.submit_batch_device(ds[0])
    for i in range(K):
        nxt = c.submit_batch_device(ds[(i + 1) % R]) if i + 1 < K else None
        rows = c.wait_batch(t, keep_results="packed")
        t = nxt
    torch.cuda.synchronize()
    print("submit/wait %.4f ms per step (%d rows)" % ((time.perf_counter() - t0) / K * 1e3, len(rows)), flush=True)
