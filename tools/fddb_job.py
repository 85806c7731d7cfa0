"""The FDDB-shaped job of bench.py (2,845 images <= 450x450 as ONE ragged job, images resident in HBM) alone, for kernel
timelines and option A/Bs:   python tools/fddb_job.py [reps] ["NAME=VALUE ..." ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
mp = bench.model_path((5, 540, 27, 4), "cascade", 1, calib)
n_img = int(os.environ.get("FDDB_N", "2845"))
rng = np.random.default_rng(0)
sizes = []
for _ in range(n_img):
    long_side = int(rng.integers(300, 451)); short = int(rng.integers(225, long_side + 1))
    sizes.append((long_side, short) if rng.random() < 0.5 else (short, long_side))
base = synth.make_frames(64, 450, 450, seed=7)
imgs = [np.ascontiguousarray(base[i % 64][:sizes[i][1], :sizes[i][0]]) for i in range(n_img)]
offs, tot = [], 0
for im in imgs:
    offs.append(tot); tot += im.size
buf = np.concatenate([im.reshape(-1) for im in imgs])
ws, hs = [s[0] for s in sizes], [s[1] for s in sizes]
d_buf = torch.from_numpy(buf).cuda()
base_env = dict(os.environ)
for spec in (sys.argv[2:] or [""]):
    os.environ.clear(); os.environ.update(base_env)
    for kv in spec.split():
        k, v = kv.split("=", 1); os.environ[k] = v
    import _dummy_streams; _dummy_streams.make()
    c = api.Cascador(mp)
    for _ in range(2):
        rows, st = c.detect_ragged_packed(d_buf, offs, ws, hs, stats=True, keep_results="packed")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        rows, st = c.detect_ragged_packed(d_buf, offs, ws, hs, stats=True, keep_results="packed")
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / reps
    print("%-60s %.3f ms per job  %.0f images/s  %.3e windows/s  gpu %.3f scan %.3f  rows %d  launches %d" %
          (spec or "(defaults)", el * 1e3, n_img / el, st["patch_n"] / el, st["gpu_ms"], st["scan_ms"], len(rows), st["scan_launches"]), flush=True)
    c.close()
