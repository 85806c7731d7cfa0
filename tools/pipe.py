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
import _dummy_streams; _dummy_streams.make()
c = api.Cascador(mp)
K = int(os.environ.get("PIPE_STEPS", "40"))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    A = int(os.environ.get("PIPE_AHEAD", "1"))         # batches submitted ahead of the one being waited for
    q = [c.submit_batch_device(ds[j % R]) for j in range(min(A, K))]
    issued = len(q)
    for i in range(K):
        if issued < K:
            q.append(c.submit_batch_device(ds[issued % R])); issued += 1
        rows = c.wait_batch(q.pop(0), keep_results="packed")
    torch.cuda.synchronize()
    print("submit/wait %.4f ms per step (%d rows)" % ((time.perf_counter() - t0) / K * 1e3, len(rows)), flush=True)
