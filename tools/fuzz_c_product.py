"""Randomised comparison of the PRODUCT's dialect C (libjda.so on the GPU: jdaDetect, jdaDetectBatch, the ragged entry) with the
REFERENCE ITSELF -- c/jda.c compiled by oracle/build.py for the dimension sets that have a build (oracle/_ref/) -- on random
models of those dimensions, frames, and call parameters (scale, min_size, max_size, th): boxes, scores and landmarks after
NMS and relocation, bit for bit.  This is the PINNED dialect.   python tools/fuzz_c_product.py [seed] [seconds]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jda_amd import api, synth
from oracle import build as obuild
from oracle.pyoracle import Reference, reference_lib_path
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60
same = lambda a, b: a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
dims_all = [d for d in obuild.REF_DIMS if reference_lib_path(*d) and d[1] <= 80]
assert dims_all, "no compiled reference under oracle/_ref"
tmp = tempfile.mkdtemp()
t0 = time.time(); n = 0; ndet = 0; calls = 0
while time.time() - t0 < secs:
    dims = dims_all[int(rng.integers(0, len(dims_all)))]
    rb = int(rng.choice([8, 4]))
    mdl = synth.make_model(*dims, seed=int(rng.integers(0, 1 << 30)), cart_th=float(rng.uniform(-3, 0.3)), norm_every=int(rng.integers(1, 7)),
                           w_sigma=float(rng.choice([2e-3, 2e-2, 1e-1])))
    if rng.integers(0, 2): mdl.off *= float(rng.uniform(1, 4))            # offsets past the window: the clamps
    p = os.path.join(tmp, "m_%d.model" % n)
    mdl.save(p, rb)
    c, ref = api.Cascador(p, "double" if rb == 8 else "float"), Reference(p, dims, rb)
    sizes = [(int(rng.integers(24, 260)), int(rng.integers(24, 200))) for _ in range(int(rng.integers(1, 6)))]
    imgs = [synth.make_frames(1, w, h, seed=int(rng.integers(0, 1 << 30)))[0] for (w, h) in sizes]
    scale = float(np.float32(rng.choice([1.1, 1.2, 1.25, 1.5, 2.0, 1.0 + rng.uniform(0.05, 1.0)])))
    min_size = int(rng.choice([0, 24, 30, 40, 64]))
    max_size = int(rng.choice([-1, 0, 60, 100, 500]))
    th = float(np.float32(rng.uniform(-2, 2)))
    ctx = (dims, rb, sizes, scale, min_size, max_size, th)
    want = [ref.detect(im, scale, min_size, max_size, th) for im in imgs]
    for im, w_ in zip(imgs, want):
        got = c.detect(im, scale, 0.1, min_size, max_size, th)
        for k in ("bboxes", "scores", "shapes"):
            assert same(got[k], w_[k]), (ctx, k, got[k].shape, w_[k].shape)
        ndet += len(w_["scores"]); calls += 1
    rag = c.detect_ragged(imgs, scale, min_size, max_size, th)
    for i, w_ in enumerate(want):
        for k in ("bboxes", "scores", "shapes"):
            assert same(rag[i][k], w_[k]), (ctx, "ragged", i, k)
    same_size = [im for im in imgs if im.shape == imgs[0].shape]
    bat = c.detect_batch(np.stack(same_size), scale, min_size, max_size, th)
    for g, im in zip(bat, same_size):
        w_ = ref.detect(im, scale, min_size, max_size, th)
        for k in ("bboxes", "scores", "shapes"):
            assert same(g[k], w_[k]), (ctx, "batch", k)
    c.close(); ref.close(); n += 1
print("models %d, jdaDetect calls %d, detections %d: product == compiled reference (c/jda.c), also through the batch and ragged entries" % (n, calls, ndet))
