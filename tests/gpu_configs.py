"""BASELINE.json configs[2] (1080p, 8 levels) and configs[4] (T=7,K=2000,L=68,D=6) on the GPU:
parity against the oracle on one frame + throughput of a batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from jda_amd import synth, api
from oracle.pyoracle import Oracle


def bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


def parity(c, o, frame, **kw):
    g = c.trace(frame[None], **kw); r = o.trace(frame, **kw)
    bad = {k: int((bits(r[k]) != bits(g[k])).sum()) for k in r}
    d = c.detect_batch(frame[None], **kw)[0]; w = o.detect(frame, **kw)
    dbad = sum(0 if (w[k].shape == d[k].shape and np.array_equal(bits(w[k]), bits(d[k]))) else 1 for k in w)
    return bad, dbad, len(d["scores"])


def throughput(c, frames, steps=3, **kw):
    d = torch.from_numpy(frames).cuda()
    c.detect_batch_device(d, keep_results=False, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        out, st = c.detect_batch_device(d, keep_results=False, stats=True, **kw)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / steps
    return st["patch_n"] / el, el * 1e3, st


which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("3", "both"):
    t0 = time.time()
    frames = synth.make_frames(32, 1920, 1080, seed=0)
    m = synth.make_model(5, 540, 27, 4, seed=1)
    synth.calibrate_thresholds(m, frames[:4], scale=1.5)
    p = "/tmp/cfg3.model"; m.save(p, 8)
    print("config 3 setup s %.1f" % (time.time() - t0), flush=True)
    c, o = api.Cascador(p), Oracle(p)
    print("config 3 parity (1 frame 1080p, scale 1.5):", parity(c, o, frames[0], scale=1.5), flush=True)
    wps, ms, st = throughput(c, frames, scale=1.5)
    print("config 3 throughput: %.3e windows/s, %.2f ms per 32-frame batch, avg carts %.1f, gpu_ms %.2f scan_ms %.2f"
          % (wps, ms, st["average_cart_n"], st["gpu_ms"], st["scan_ms"]), flush=True)
if which in ("5", "both"):
    t0 = time.time()
    m = synth.make_model(7, 2000, 68, 6, seed=2, cart_th=-2.0)
    p = "/tmp/cfg5.model"; m.save(p, 4)
    print("config 5 setup s %.1f, file MB %.1f" % (time.time() - t0, os.path.getsize(p) / 1e6), flush=True)
    c, o = api.Cascador(p, "float"), Oracle(p)
    small = synth.make_frames(1, 320, 240, seed=3)[0]
    print("config 5 parity (320x240):", parity(c, o, small), flush=True)
    frames = synth.make_frames(4, 1920, 1080, seed=4)
    wps, ms, st = throughput(c, frames, steps=2)
    print("config 5 throughput 1080p canonical: %.3e windows/s, %.2f ms per 4-frame batch, avg carts %.1f, stage_done %s"
          % (wps, ms, st["average_cart_n"], st["stage_done_n"][:7]), flush=True)
