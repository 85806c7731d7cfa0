"""All-pass regime timing (every window walks all T*K carts): gpu ms per batch for a few env settings.
usage: python tools/allpass.py [--dims 5,540,27,4] [--batch 64] [--steps 2] [VAR=v ...]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from jda_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--dims", default="5,540,27,4")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--cart-th", type=float, default=None)
ap.add_argument("env", nargs="*")
a = ap.parse_args()
for kv in a.env:
    k, v = kv.split("=")
    os.environ[k] = v
dims = tuple(int(x) for x in a.dims.split(","))
mp = "/tmp/allpass_%s_%s.model" % ("_".join(map(str, dims)), a.cart_th)
kw = {} if a.cart_th is None else {"cart_th": a.cart_th}
synth.make_model(*dims, seed=1, **kw).save(mp, 4)
frames = synth.make_frames(a.batch, 640, 480, seed=7)
d = torch.from_numpy(frames).cuda()
c = api.Cascador(mp)
for i in range(a.steps + 1):
    t0 = time.perf_counter()
    _, st = c.detect_batch_device(d, 1.25, 40, -1, float("inf"), nms=True, stats=True, keep_results="packed")
    el = (time.perf_counter() - t0) * 1e3
    if i:
        print("%s dims %s batch %d: gpu_ms %.2f call_ms %.2f dense %d avg_carts %.1f carts/s %.3e win/s %.3e" % (
            " ".join(a.env), dims, a.batch, st["gpu_ms"], el, st["dense_passes"], st["cart_total_n"] / st["patch_n"],
            st["cart_total_n"] / (st["gpu_ms"] * 1e-3), st["patch_n"] / (st["gpu_ms"] * 1e-3)))
