"""Dialect CPP tied to the PINNED dialect C on a model where the two must agree (CPU part).

`src/jda` (fp64, round(), cv::Mat) cannot be compiled in this image, so the restatement of dialect CPP in
oracle/jda_oracle.c has no reference-held pin.  This file narrows what that leaves open: on a model whose reals
are small dyadic rationals (jda_amd/synth.py:make_dyadic_model) truncation and round() select the same pixels and
fp32 and fp64 compute every score and shape coordinate exactly, so

    compiled reference c/jda.c  ==  oracle dialect C   (final detections, bit for bit: the existing pin)
    oracle dialect C trace      ==  oracle dialect CPP trace   (per window: carts evaluated = reject position,
                                                                 leaf-path hash, score, shape)

must both hold -- the second line is the new one: it runs Validate / Cart::Forward / CalcFeatureValue /
GenDeltaShape as restated from cascador.cpp:166-211, cart.cpp:392-404, data.cpp:18-58, btcart.cpp:407-424 against
the walk of c/jda.c:357-411 that the reference itself pins.  What stays unpinned in dialect CPP after this:
round() vs truncation where they differ, fp64 accumulation where fp32 rounds, the patch sizes of scale != 0 nodes,
cv::resize, the multimap NMS.  The GPU half (the kernels' fp64 instantiation against their pinned fp32 one) is
tests/test_gpu_parity.py::test_dialects_agree_where_they_must.
"""
import numpy as np
import pytest

from conftest import same

CASES = [  # dims (a reference build of these dims exists: oracle/build.py DIMS), window side, C scale that reaches it from 24
    ((3, 20, 5, 4), 32, 4.0 / 3.0),
    ((3, 70, 9, 5), 64, 8.0 / 3.0),
    ((2, 64, 68, 6), 32, 4.0 / 3.0),
]


def dialect_calls(win, scale):
    """The two calls that enumerate the SAME window grid: one window size, step (int)(win * 0.1f)."""
    step = int(np.float32(win) * np.float32(0.1))
    return dict(scale=np.float32(scale), min_size=win, max_size=win), dict(minimum_size=win, step=step, factor=100.0)


@pytest.mark.parametrize("dims,win,scale", CASES)
@pytest.mark.parametrize("reject", [0.0, 0.12])
def test_oracle_dialects_walk_the_same_path(tmp_path, dims, win, scale, reject):
    from jda_amd import synth
    from oracle.pyoracle import Oracle
    m = synth.make_dyadic_model(*dims, win=win, seed=11 + win, reject=reject)
    p = str(tmp_path / "dyadic.model")
    m.save(p, 8)
    kc, kp = dialect_calls(win, scale)
    o = Oracle(p)
    frame = synth.make_frames(1, 150, 110, seed=5)[0]
    a = o.trace(frame, **kc)
    b = o.trace_cpp(frame, **kp)
    n = len(a["carts_n"])
    assert n == len(b["carts_n"]) == ((150 - win) // kp["step"] + 1) * ((110 - win) // kp["step"] + 1)
    assert np.array_equal(a["carts_n"], b["carts_n"])                      # reject position (Validate's n)
    assert np.array_equal(a["path_hash"], b["path_hash"])                  # leaf index of every evaluated cart
    assert same(a["score"], b["score"].astype(np.float32))                 # the fp64 score IS an fp32 number here
    assert np.array_equal(b["score"].astype(np.float32).astype(np.float64), b["score"])
    assert same(a["shapes"], b["shapes"].astype(np.float32))
    assert np.array_equal(b["shapes"].astype(np.float32).astype(np.float64), b["shapes"])
    T, K = dims[0], dims[1]
    if reject == 0.0:
        assert (a["carts_n"] == T * K).all()
    else:
        died = a["carts_n"] < T * K
        assert 0.2 < died.mean() and len(np.unique(a["carts_n"])) > 10     # a real cascade: many reject positions ...
        assert (a["carts_n"] > K).any()                                    # ... and windows that ran the regression
    assert len(np.unique(a["path_hash"])) > n // 4                         # the walks really depend on the pixels


@pytest.mark.parametrize("dims,win,scale", CASES)
def test_the_dyadic_model_stays_under_the_reference_pin(tmp_path, dims, win, scale):
    """The model of the differential test, through the reference's own compiled c/jda.c: the C side of the comparison
    above is the pinned one on THIS model too, not only on the random ones of test_oracle_vs_reference.py."""
    from jda_amd import synth
    from oracle.pyoracle import Oracle, Reference, reference_lib_path
    if reference_lib_path(*dims) is None:
        pytest.skip("no reference build for %s" % (dims,))
    m = synth.make_dyadic_model(*dims, win=win, seed=11 + win, reject=0.12)
    p = str(tmp_path / "dyadic.model")
    m.save(p, 8)
    kc, _ = dialect_calls(win, scale)
    frame = synth.make_frames(1, 150, 110, seed=5)[0]
    ref, orc = Reference(p, dims, 8), Oracle(p)
    th = -1.0
    want = ref.detect_raw(frame, float(kc["scale"]), win, win, th)
    tr = orc.trace(frame, **kc)
    T, K = dims[0], dims[1]
    alive = (tr["carts_n"] == T * K) & ~(tr["score"] < np.float32(th))
    assert alive.sum() == len(want["scores"]) > 0
    assert same(want["scores"], tr["score"][alive]) and same(want["shapes"], tr["shapes"][alive])
    got = orc.detect(frame, float(kc["scale"]), win, win, th)
    full = ref.detect(frame, float(kc["scale"]), win, win, th)
    for k in full:
        assert same(full[k], got[k]), k
