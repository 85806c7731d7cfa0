#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Run in a container that has /root/reference:  python tests/golden/make_golden.py

For every case below it builds a synthetic model + frame (jda_amd.synth, seeded),
runs the reference's own c/jda.c -- compiled by oracle/build.py from where it lies
under /root/reference, dimension #defines rewritten on the fly -- and stores:

  inputs : the frame, the call arguments, the model (its bytes when small, else
           the generator arguments + sha256 of the bytes)
  outputs: jdaDetect (post-NMS, relocated), jdaInternalDetect (pre-NMS survivors,
           via ref_detect_raw), jdaImageResize half/quarter images (ref_resize)

The fixtures are data only (inputs + expected outputs); no reference source is
stored.  tests/test_golden.py checks the oracle against them on CPU and
tests/test_gpu_parity.py checks the HIP path against them on the GPU box.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from jda_amd import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402

CASES = [
    # name, dims, model kwargs, frame (w,h,seed|'const'), call (scale,min,max,th)
    ("tiny_a", (2, 8, 5, 3), dict(seed=3, cart_th=-0.3, norm_every=3, f32_exact=False), (96, 80, 5), (1.25, 24, -1, -0.5)),
    ("tiny_b", (3, 20, 5, 4), dict(seed=4, cart_th=-1.0, norm_every=5), (120, 90, 6), (1.25, 40, -1, -0.5)),
    ("deep", (2, 6, 4, 6), dict(seed=5, cart_th=-0.5, norm_every=4), (100, 100, 7), (1.5, 30, -1, 0.0)),
    ("stump", (1, 4, 3, 2), dict(seed=6), (64, 48, 8), (1.25, 24, 40, -10.0)),
    ("odd_width", (3, 70, 9, 5), dict(seed=7, cart_th=-1.0, norm_every=20), (131, 97, 9), (1.2, 30, -1, -0.5)),
    ("wide_shape", (2, 64, 68, 6), dict(seed=8, cart_th=-1.0, norm_every=16), (110, 90, 10), (1.25, 40, -1, -0.5)),
    ("one_window", (2, 8, 5, 3), dict(seed=3), (24, 24, 11), (1.25, 0, -1, -10.0)),
    ("no_window", (2, 8, 5, 3), dict(seed=3), (23, 30, 12), (1.25, 40, -1, -10.0)),
    ("score_ties", (2, 8, 5, 3), dict(seed=3), (80, 60, "const"), (1.25, 24, -1, -10.0)),
    ("shipped_allpass", (5, 540, 27, 4), dict(seed=1), (100, 75, 13), (1.25, 40, -1, -10.0)),
    ("shipped_cut", (5, 540, 27, 4), dict(seed=1, cart_th=-2.0), (200, 150, 14), (1.25, 40, -1, -0.5)),
]
INLINE_MODEL_LIMIT = 70000


def frame_of(spec):
    w, h, s = spec
    if s == "const":
        return np.full((h, w), 128, np.uint8)
    return synth.make_frames(1, w, h, seed=s)[0]


def main():
    index = {}
    for name, dims, mkw, fspec, call in CASES:
        model = synth.make_model(*dims, **mkw)
        blob = model.tobytes(8)
        mp = "/tmp/golden_%s.model" % name
        open(mp, "wb").write(blob)
        frame = frame_of(fspec)
        ref = pyoracle.Reference(mp, dims, 8)
        scale, mn, mx, th = call
        post = ref.detect(frame, scale, mn, mx, th)
        raw = ref.detect_raw(frame, scale, mn, mx, th)
        h, w = frame.shape
        r = np.float32(1.0) / np.sqrt(np.float32(2.0))
        hw, hh, qw, qh = int(np.float32(w) * r), int(np.float32(h) * r), w // 2, h // 2
        half = ref.resize(frame, hw, hh) if hw > 0 and hh > 0 else np.zeros((0, 0), np.uint8)
        quarter = ref.resize(frame, qw, qh) if qw > 0 and qh > 0 else np.zeros((0, 0), np.uint8)
        arrays = dict(frame=frame, post_bboxes=post["bboxes"], post_scores=post["scores"], post_shapes=post["shapes"],
                      raw_bboxes=raw["bboxes"], raw_scores=raw["scores"], raw_shapes=raw["shapes"],
                      half=half, quarter=quarter)
        if len(blob) <= INLINE_MODEL_LIMIT:
            arrays["model_bytes"] = np.frombuffer(blob, np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
        index[name] = dict(dims=list(dims), model_kwargs=mkw, model_sha256=hashlib.sha256(blob).hexdigest(),
                           model_inline=len(blob) <= INLINE_MODEL_LIMIT, frame=list(fspec), call=list(call),
                           n_post=int(len(post["scores"])), n_raw=int(len(raw["scores"])))
        print(name, "raw", len(raw["scores"]), "post", len(post["scores"]))
    json.dump(index, open(os.path.join(HERE, "index.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
