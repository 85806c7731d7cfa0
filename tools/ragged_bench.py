#!/usr/bin/env python3
"""BASELINE.json configs[3] as ONE ragged job: 2,845 FDDB-sized synthetic images (<= 450x450, varied aspect), shipped
model dimensions, cascade regime, canonical call -- through jdaDetectBatchRagged (host images: separate allocations /
one packed buffer, pageable or pinned) and jdaDetectBatchRaggedDevice (images resident in HBM), from one host thread.
Prints one JSON line per variant."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def fddb_sizes(n, seed=0, max_side=450):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        long_side = int(rng.integers(max_side * 2 // 3, max_side + 1))
        short = int(rng.integers(max_side // 2, long_side + 1))
        out.append((long_side, short) if rng.random() < 0.5 else (short, long_side))
    return out


def fddb_model():
    from jda_amd import synth
    mp = os.path.join(synth.cache_dir(), "fddb_model_c.model")
    if not os.path.exists(mp):
        m = synth.make_model(5, 540, 27, 4, seed=1)
        synth.calibrate_thresholds(m, synth.make_frames(8, 450, 450, seed=0, first=10_000_000), scale=1.25, min_size=40)
        m.save(mp + ".%d" % os.getpid(), 8); os.replace(mp + ".%d" % os.getpid(), mp)
    return mp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2845)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--variants", default="device,packed,packed_pinned,list")
    args = ap.parse_args()
    import torch
    from jda_amd import api, synth
    sizes = fddb_sizes(args.images)
    imgs = [synth.make_frames(1, w, h, seed=1, first=i)[0] for i, (w, h) in enumerate(sizes)]
    offs, tot = [], 0
    for im in imgs:
        offs.append(tot); tot += im.size
    buf = np.concatenate([im.reshape(-1) for im in imgs])
    ws, hs = [s[0] for s in sizes], [s[1] for s in sizes]
    c = api.Cascador(fddb_model())
    dev = torch.device("cuda", 0)
    d_buf = torch.from_numpy(buf).to(dev)
    pin = torch.from_numpy(buf).pin_memory()
    runs = {
        "device": lambda: c.detect_ragged_packed(d_buf, offs, ws, hs, stats=True, keep_results="packed"),
        "packed": lambda: c.detect_ragged_packed(buf, offs, ws, hs, stats=True, keep_results="packed"),
        "packed_pinned": lambda: c.detect_ragged_packed(pin.numpy(), offs, ws, hs, stats=True, keep_results="packed"),
        "list": lambda: c.detect_ragged(imgs, stats=True, keep_results="packed"),
    }
    for name in args.variants.split(","):
        fn = runs[name]
        fn(); fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter(); rows, st = fn(); ts.append(time.perf_counter() - t0)
        el = float(np.median(ts))
        print(json.dumps({"metric": "FDDB-shaped images/sec, one ragged job from one host thread", "variant": name,
                          "value": args.images / el, "unit": "images/s", "images": args.images, "ms_per_job": el * 1e3,
                          "best_ms": min(ts) * 1e3, "windows": st["patch_n"], "windows_per_s": st["patch_n"] / el,
                          "gpu_ms_sum": st["gpu_ms"], "host_post_ms": st["host_ms"], "detections": int(len(rows)),
                          "bytes": int(tot), "average_cart_n": st["average_cart_n"]}), flush=True)


if __name__ == "__main__":
    main()
