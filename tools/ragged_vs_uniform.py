"""The same 256 frames 640x480 through the uniform batch entry and through the ragged entry (one job): what the ragged
machinery itself costs (block map, per-image segments, repack, chunks) when the geometry is identical."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
import bench
mp = bench.model_path((5, 540, 27, 4), "cascade", 1, synth.make_frames(8, 640, 480, seed=0, first=10_000_000))
frames = synth.make_frames(256, 640, 480, seed=0)
d = torch.from_numpy(frames).cuda()
c = api.Cascador(mp)
for cw in (int(x) for x in (sys.argv[1:] or ["6000000"])):
    c.set_option("ragged_chunk_windows", cw)
    offs = [i * 640 * 480 for i in range(256)]; ws = [640] * 256; hs = [480] * 256
    for _ in range(3):
        c.detect_batch_device(d, keep_results=False); c.detect_ragged_packed(d.view(-1), offs, ws, hs, keep_results=False)
    for name, fn in (("uniform", lambda: c.detect_batch_device(d, keep_results=False, stats=True)),
                     ("ragged ", lambda: c.detect_ragged_packed(d.view(-1), offs, ws, hs, keep_results=False, stats=True))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            _, st = fn()
        el = (time.perf_counter() - t0) / 10
        print("chunk %9d  %s %.3f ms per 256 frames  gpu_ms %.3f scan_ms %.3f  %.3e windows/s" % (cw, name, el * 1e3, st["gpu_ms"], st["scan_ms"], st["patch_n"] / el), flush=True)
