"""Randomised comparison of the PRODUCT's dialect CPP (libjda.so on the GPU: jdaTraceBatchCpp, jdaDetectBatchCpp, the ragged entry)
with the oracle's restatement: random model shapes incl. trainer snapshots and multi-scale models, images, scan parameters, the
similarity transform -- per window and per image, bit for bit.   python tools/fuzz_cpp_product.py [seed] [seconds]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jda_amd import api, synth
from oracle.pyoracle import Oracle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60
same = lambda a, b: a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
tmp = tempfile.mkdtemp()
t0 = time.time(); n = 0; nwin = 0; snaps = 0
while time.time() - t0 < secs:
    T, K, L, D = int(rng.integers(1, 4)), int(rng.integers(2, 40)), int(rng.integers(2, 10)), int(rng.integers(2, 6))
    multi = bool(rng.integers(0, 3) == 0)
    mdl = synth.make_model(T, K, L, D, seed=int(rng.integers(0, 1 << 30)), cart_th=float(rng.uniform(-3, 0.2)), multi_scale=multi,
                           norm_every=int(rng.integers(1, 6)), w_sigma=float(rng.choice([2e-3, 2e-2, 1e-1])))
    if rng.integers(0, 2): mdl.off *= float(rng.uniform(1, 4))
    sim = bool(rng.integers(0, 3) == 0)
    hdr = None
    if rng.integers(0, 2) == 0: hdr = (int(rng.integers(0, T)), int(rng.integers(-1, K)))
    p = os.path.join(tmp, "m_%d.model" % n)
    mdl.save(p, 8, **({} if hdr is None else dict(header_stage=hdr[0], header_cart=hdr[1])))
    sizes = [(int(rng.integers(20, 160)), int(rng.integers(20, 120))) for _ in range(int(rng.integers(1, 5)))]
    imgs = [synth.make_frames(1, w, h, seed=int(rng.integers(0, 1 << 30)))[0] for (w, h) in sizes]
    ms, st, fa = int(rng.integers(12, 30)), int(rng.integers(2, 9)), float(rng.choice([1.1, 1.2, 1.5, 2.0]))
    ctx = (T, K, L, D, multi, hdr, sizes, ms, st, fa, sim)
    c, o = api.Cascador(p), Oracle(p)
    c.set_similarity_transform(sim); o.set_similarity_transform(sim)
    try:
        ran = None if hdr is None else hdr[0] * K + hdr[1] + 1
        for im in imgs:
            got, want = c.trace_cpp(im[None], ms, st, fa), o.trace_cpp(im, ms, st, fa)
            face = got["carts_n"] == T * K
            if ran is not None and ran < T * K:
                assert (want["carts_n"][face] == ran).all(), ctx
            else:
                assert np.array_equal(got["carts_n"][face], want["carts_n"][face]), ctx
            assert np.array_equal(got["carts_n"][~face], want["carts_n"][~face]), ctx
            assert np.array_equal(got["path_hash"][~face], want["path_hash"][~face]), ctx
            assert same(got["score"], want["score"]) and same(got["shapes"], want["shapes"]), ctx
            nwin += len(face)
        nms = bool(rng.integers(0, 2))
        rag = c.detect_ragged_cpp(imgs, ms, st, fa, 0.3, nms)
        for i, im in enumerate(imgs):
            want = o.detect_cpp(im, ms, st, fa, 0.3, nms)
            one = c.detect_batch_cpp(im[None], ms, st, fa, 0.3, nms)[0]
            for k in ("rects", "scores", "shapes"):
                assert same(rag[i][k], want[k]) and same(one[k], want[k]), (ctx, i, k)
    finally:
        o.set_similarity_transform(False)
    c.close(); o.close(); n += 1; snaps += hdr is not None
print("cases %d (%d trainer snapshots), windows %d: product == oracle" % (n, snaps, nwin))
