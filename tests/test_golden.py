"""The oracle (oracle/jda_oracle.c) against golden vectors produced by the
reference's own compiled c/jda.c (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import golden_util
from conftest import same


@pytest.mark.parametrize("name", golden_util.NAMES)
def test_oracle_reproduces_reference_outputs(built, tmp_path, name):
    from oracle.pyoracle import Oracle
    meta, g, mp = golden_util.load(name, tmp_path)
    scale, mn, mx, th = meta["call"]
    o = Oracle(mp)
    post = o.detect(g["frame"], scale, mn, mx, th, nms=True)
    raw = o.detect(g["frame"], scale, mn, mx, th, nms=False)
    assert len(raw["scores"]) == meta["n_raw"] and len(post["scores"]) == meta["n_post"]
    for k in ("bboxes", "scores", "shapes"):
        assert same(raw[k], g["raw_" + k]), (name, "raw", k)     # bit-exact, pre-NMS
        assert same(post[k], g["post_" + k]), (name, "post", k)  # bit-exact, post-NMS + relocation
    # trace is consistent with the detect outputs: survivors = windows that walked T*K carts
    T, K = meta["dims"][0], meta["dims"][1]
    tr = o.trace(g["frame"], scale, mn, mx)
    passed = (tr["carts_n"] == T * K) & ~(tr["score"] < np.float32(th))
    # (a window may also be rejected exactly at the last cart; those have carts_n == T*K too)
    assert passed.sum() >= meta["n_raw"]
    sel = np.flatnonzero(passed)
    got = {tuple(r) for r in tr["shapes"][sel].view(np.uint32).tolist()}
    for row in g["raw_shapes"].view(np.uint32).tolist():
        assert tuple(row) in got


@pytest.mark.parametrize("name", golden_util.NAMES)
def test_oracle_resize(built, tmp_path, name):
    from oracle.pyoracle import Oracle
    meta, g, mp = golden_util.load(name, tmp_path)
    o = Oracle(mp)
    h, w = g["frame"].shape
    hw, hh, qw, qh = o.pyramid_dims(w, h)
    assert g["half"].shape == ((hh, hw) if hw > 0 and hh > 0 else (0, 0))
    if hw > 0 and hh > 0:
        assert np.array_equal(o.resize(g["frame"], hw, hh), g["half"])
    if qw > 0 and qh > 0:
        assert np.array_equal(o.resize(g["frame"], qw, qh), g["quarter"])
