"""The bench's headline leg (submit/wait, PIPE_AHEAD batches queued ahead, four rotating resident batches) under several
settings of the JDA_* knobs, one process:   python tools/pipe_variants.py "NAME=VALUE ..." "" ...
Every variant gets a fresh cascador; results are checked against the first variant's (row count and checksum)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
import bench
calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
mp = bench.model_path((5, 540, 27, 4), "cascade", 1, calib)
R = 4
ds = [torch.from_numpy(synth.make_frames(256, 640, 480, seed=0, first=i * 256)).cuda() for i in range(R)]
K = int(os.environ.get("PIPE_STEPS", "60"))
A = int(os.environ.get("PIPE_AHEAD", "2"))
base_env = dict(os.environ)
ref = None
for spec in (sys.argv[1:] or [""]):
    os.environ.clear(); os.environ.update(base_env)
    for kv in spec.split():
        k, v = kv.split("=", 1); os.environ[k] = v
    c = api.Cascador(mp)
    best, sums = 1e9, None
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q = [c.submit_batch_device(ds[j % R]) for j in range(min(A, K))]
        issued = len(q)
        acc = 0.0
        for i in range(K):
            if issued < K:
                q.append(c.submit_batch_device(ds[issued % R])); issued += 1
            rows = c.wait_batch(q.pop(0), keep_results="packed")
            acc += float(np.asarray(rows, np.float64).sum()) + len(rows)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / K * 1e3
        if rep: best = min(best, el)
        sums = acc
    if ref is None: ref = sums
    print("%-70s %.4f ms per step  %s" % (spec or "(defaults)", best, "same" if sums == ref else "DIFFERENT RESULTS"), flush=True)
    c.close()
