"""Randomised differential test of the call surface: frame sizes from 1x1 up, scales, size limits and thresholds drawn
at random (fixed seed) through jdaDetect, jdaDetectBatch, jdaDetectBatchRagged and jdaDetectBatchCpp, every image compared
bit for bit with the oracle (c/jda.c:318-480 / cascador.cpp:310-477 restated).  The named edge cases live in
test_gpu_parity.py / test_ragged.py; this one looks for the combinations nobody named.  It is also what
tools/sessions/r05_w.sh runs on the sanitizer build of the host side."""
import numpy as np
import pytest

from conftest import same

pytestmark = pytest.mark.gpu

MODELS = [((3, 20, 5, 4), dict(seed=3, cart_th=-1.0, norm_every=5)),
          ((2, 8, 5, 3), dict(seed=5, cart_th=-0.3, norm_every=3)),
          ((3, 70, 9, 5), dict(seed=7, cart_th=-0.6, norm_every=7))]


def _eq(a, b, what):
    for k in a:
        if k in b:
            assert same(a[k], b[k]), (what, k, np.asarray(a[k]).shape, np.asarray(b[k]).shape)


def _relocated(raw):
    """the oracle's nms=False output is jdaInternalDetect's (normalised shapes); the library's is relocated: x*size then +origin"""
    out = dict(raw)
    sz = raw["bboxes"][:, 2].astype(np.float32)[:, None]
    sh = raw["shapes"].copy()
    sh[:, 0::2] = sh[:, 0::2] * sz + raw["bboxes"][:, 0].astype(np.float32)[:, None]      # two roundings, c/jda.c:471-472
    sh[:, 1::2] = sh[:, 1::2] * sz + raw["bboxes"][:, 1].astype(np.float32)[:, None]
    out["shapes"] = sh
    return out


def _size(rng):
    kind = rng.integers(5)
    if kind == 0:
        return int(rng.integers(1, 30)), int(rng.integers(1, 30))             # below / around the smallest window
    if kind == 1:
        return int(rng.integers(24, 60)), int(rng.integers(100, 320))         # narrow
    if kind == 2:
        return int(rng.integers(100, 320)), int(rng.integers(24, 60))         # flat
    return int(rng.integers(40, 330)), int(rng.integers(40, 330))


def _img(rng, w, h):
    from jda_amd import synth
    if rng.integers(6) == 0:
        return np.full((h, w), int(rng.integers(256)), np.uint8)              # flat image: ties everywhere
    return synth.make_frames(1, w, h, seed=int(rng.integers(1 << 20)))[0]


@pytest.mark.parametrize("mi", range(len(MODELS)))
def test_random_calls_dialect_c(built, model_file, mi):
    import torch
    assert torch.cuda.is_available()
    from jda_amd import api
    from oracle.pyoracle import Oracle
    dims, kw = MODELS[mi]
    p, _ = model_file(dims, 8, **kw)
    c, o = api.Cascador(p), Oracle(p)
    rng = np.random.default_rng(100 + mi)
    n_det = n_img = 0
    for it in range(40):
        call = dict(scale=float(rng.choice([1.05, 1.1, 1.25, 1.3, 1.5, 2.0, 3.7])), min_size=int(rng.choice([-3, 0, 24, 25, 40, 41, 77, 150])),
                    max_size=int(rng.choice([-1, 0, 23, 24, 30, 60, 100, 1000])), th=float(rng.choice([-5.0, -0.5, 0.0, 0.4, 3.0])))
        entry = rng.integers(3)
        if entry == 0:                                   # the drop-in call
            w, h = _size(rng)
            im = _img(rng, w, h)
            _eq(c.detect(im, **call), o.detect(im, **call), ("detect", it, w, h, call))
            n_img += 1
        elif entry == 1:                                 # a batch of equal frames, with and without NMS
            w, h = _size(rng)
            nms = bool(rng.integers(2))
            fr = np.stack([_img(rng, w, h) for _ in range(int(rng.integers(1, 6)))])
            got = c.detect_batch(fr, nms=nms, **call)
            for i in range(len(fr)):
                want = o.detect(fr[i], nms=nms, **call)
                if not nms:
                    want = _relocated(want)
                _eq(got[i], want, ("batch", it, i, w, h, nms, call))
                n_det += len(want["scores"])
            n_img += len(fr)
        else:                                            # a ragged list
            sizes = [_size(rng) for _ in range(int(rng.integers(1, 10)))]
            ims = [_img(rng, w, h) for w, h in sizes]
            got = c.detect_ragged(ims, **call)
            for i, im in enumerate(ims):
                want = o.detect(im, **call)
                _eq(got[i], want, ("ragged", it, i, sizes[i], call))
                n_det += len(want["scores"])
            n_img += len(ims)
    assert n_img > 60 and n_det > 0
    c.close()


def test_random_calls_dialect_cpp(built, model_file):
    import torch
    assert torch.cuda.is_available()
    from jda_amd import api
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-1.0, norm_every=5)
    c, o = api.Cascador(p), Oracle(p)
    rng = np.random.default_rng(9)
    n_det = 0
    for it in range(25):
        w, h = _size(rng)
        call = dict(minimum_size=int(rng.choice([1, 5, 20, 21, 33, 60])), step=int(rng.choice([1, 2, 5, 9, 40])),
                    factor=float(rng.choice([1.05, 1.2, 1.5, 2.5])), overlap=float(rng.choice([0.0, 0.3, 0.9])), nms=bool(rng.integers(2)))
        if call["step"] == 1 and w * h > 150 * 150:
            call["step"] = 3                                      # (keeps the oracle's share of the run short)
        fr = np.stack([_img(rng, w, h) for _ in range(int(rng.integers(1, 4)))])
        try:
            o.detect_cpp(fr[0], **call)
        except ValueError:                                        # a factor that does not grow the window: the reference's loop
            with pytest.raises(api.JdaError):                     # (cascador.cpp:336-375) would not end; the library says so
                c.detect_batch_cpp(fr, **call)
            continue
        got = c.detect_batch_cpp(fr, **call)
        for i in range(len(fr)):
            want = o.detect_cpp(fr[i], **call)
            _eq(got[i], want, ("cpp", it, i, w, h, call))
            n_det += len(want["scores"])
    assert n_det > 0
    c.close()
