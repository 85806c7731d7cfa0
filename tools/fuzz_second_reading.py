"""Randomised comparison of the two independent restatements of the reference's fp64 detect path (oracle/jda_oracle.c and
oracle/cpp_reading2.py): random model shapes (single- and multi-scale, trainer snapshots, offsets pushed past the patch border),
images (textured, noise, four grey levels), scan parameters, the similarity transform off and on -- per window (reject length,
leaf path, score, shape) and per image (rects, scores, shapes), bit for bit.  TEST INFRASTRUCTURE; CPU only.
   python tools/fuzz_second_reading.py [seed] [seconds]        (r06: six seeds x 60 s = 17,150 cases, 2.97 M windows, no difference)"""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jda_amd import synth
from oracle import cpp_reading2 as r2
from oracle.pyoracle import Oracle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bits = lambda a: np.ascontiguousarray(a, np.float64).view(np.uint64)
t0 = time.time(); n = 0; nwin = 0
tmp = tempfile.mkdtemp()
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 60:
    T, K, L, D = int(rng.integers(1, 4)), int(rng.integers(2, 24)), int(rng.integers(2, 10)), int(rng.integers(2, 6))
    multi = bool(rng.integers(0, 2))
    mdl = synth.make_model(T, K, L, D, seed=int(rng.integers(0, 1 << 30)), cart_th=float(rng.uniform(-3, 0.2)), multi_scale=multi,
                           norm_every=int(rng.integers(1, 6)), w_sigma=float(rng.choice([2e-3, 2e-2, 1e-1])), f32_exact=bool(rng.integers(0, 2)))
    # push landmarks / offsets towards the patch border so that clamps and negative coordinates happen
    if rng.integers(0, 2): mdl.off *= float(rng.uniform(1, 4))
    hdr = None
    if rng.integers(0, 3) == 0: hdr = (int(rng.integers(0, T)), int(rng.integers(-1, K)))
    p = os.path.join(tmp, "m.model")
    mdl.save(p, 8, **({} if hdr is None else dict(header_stage=hdr[0], header_cart=hdr[1])))
    w, h = int(rng.integers(20, 70)), int(rng.integers(20, 70))
    kind = int(rng.integers(0, 3))
    img = synth.make_frames(1, w, h, seed=int(rng.integers(0, 1 << 30)))[0] if kind else rng.integers(0, 256, (h, w)).astype(np.uint8)
    if kind == 2: img = (img // 64 * 64).astype(np.uint8)       # few grey levels: feature values at the node thresholds
    ms, st, fa = int(rng.integers(12, 30)), int(rng.integers(2, 9)), float(rng.choice([1.1, 1.2, 1.5, 2.0]))
    sim = bool(rng.integers(0, 2))
    orc = Oracle(p); orc.set_similarity_transform(sim)
    try:
        tr = orc.trace_cpp(img, ms, st, fa)
        want = orc.detect_cpp(img, ms, st, fa, 0.3, True)
    finally:
        orc.set_similarity_transform(False)
    m = r2.Model2(p)
    mine = []
    rects, scores, shapes = r2.detect(m, img.tolist(), ms, st, fa, 0.3, True, resize=r2.resize_cv2 if multi else None, trace=mine, similarity=sim)
    ctx = (T, K, L, D, multi, hdr, w, h, ms, st, fa, sim, kind)
    assert len(mine) == len(tr["carts_n"]), ctx
    for i, (ok, score, shape, nn, hh) in enumerate(mine):
        assert nn == tr["carts_n"][i] and hh == tr["path_hash"][i], (ctx, i, nn, tr["carts_n"][i])
        assert bits([score])[0] == bits(tr["score"][i:i + 1])[0], (ctx, i, score, tr["score"][i])
        assert np.array_equal(bits(shape), bits(tr["shapes"][i])), (ctx, i)
    assert np.array_equal(np.array(rects, np.int32).reshape(-1, 4), want["rects"]), ctx
    assert np.array_equal(bits(scores), bits(want["scores"])) and np.array_equal(bits(np.array(shapes, np.float64).reshape(len(scores), 2 * L)), bits(want["shapes"])), ctx
    orc.close(); n += 1; nwin += len(mine)
print("cases %d, windows %d: the two readings agree" % (n, nwin))
