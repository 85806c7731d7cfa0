"""Host NMS + relocation of libjda (no GPU): exact order semantics of reference
c/jda.c:237-316 and src/jda/cascador.cpp:387-429."""
import numpy as np
import pytest

import golden_util
from conftest import same


def _relocate(shapes, bboxes):
    out = shapes.copy()
    sz = bboxes[:, 2].astype(np.float32)[:, None]
    out[:, 0::2] = out[:, 0::2] * sz + bboxes[:, 0].astype(np.float32)[:, None]   # two roundings, c/jda.c:471-472
    out[:, 1::2] = out[:, 1::2] * sz + bboxes[:, 1].astype(np.float32)[:, None]
    return out


@pytest.mark.parametrize("name", golden_util.NAMES)
def test_nms_turns_reference_raw_into_reference_post(built, tmp_path, name):
    """jdaNmsC applied to the reference's pre-NMS survivors gives the reference's post-NMS list."""
    from jda_amd import api
    meta, g, _ = golden_util.load(name, tmp_path)
    keep = api.nms_c(g["raw_bboxes"], g["raw_scores"], 0.3)
    assert same(g["raw_bboxes"][keep], g["post_bboxes"])
    assert same(g["raw_scores"][keep], g["post_scores"])
    assert same(_relocate(g["raw_shapes"][keep], g["raw_bboxes"][keep]), g["post_shapes"])


def test_ties_replay_the_exchange_sort(built):
    """Equal scores: the survivor set depends on the exchange sort's permutation (c/jda.c:256-264)."""
    from jda_amd import api
    rng = np.random.default_rng(0)
    for trial in range(30):
        n = int(rng.integers(2, 60))
        bb = np.c_[rng.integers(0, 40, n), rng.integers(0, 40, n), rng.integers(20, 40, n)].astype(np.int32)
        sc = rng.integers(0, 4, n).astype(np.float32)          # many ties
        # literal restatement of the reference loop in Python
        idx = list(range(n))
        for i in range(n - 1):
            for j in range(i + 1, n):
                if sc[idx[i]] < sc[idx[j]]:
                    idx[i], idx[j] = idx[j], idx[i]
        keep = np.ones(n, bool)
        for i in range(n - 1):
            a = idx[i]
            if not keep[a]:
                continue
            for j in range(i + 1, n):
                b = idx[j]
                if not keep[b]:
                    continue
                x1, y1 = max(bb[a, 0], bb[b, 0]), max(bb[a, 1], bb[b, 1])
                x2 = min(bb[a, 0] + bb[a, 2], bb[b, 0] + bb[b, 2]); y2 = min(bb[a, 1] + bb[a, 2], bb[b, 1] + bb[b, 2])
                w, h = max(0, x2 - x1), max(0, y2 - y1)
                ov = np.float32(w * h) / np.float32(bb[a, 2] ** 2 + bb[b, 2] ** 2 - w * h)
                if ov > np.float32(0.3):
                    keep[b] = False
        assert np.array_equal(api.nms_c(bb, sc, 0.3), keep), trial


def test_nms_edge_cases(built):
    from jda_amd import api
    assert api.nms_c(np.zeros((0, 3), np.int32), np.zeros(0, np.float32)).shape == (0,)
    assert api.nms_c([[0, 0, 30]], [1.0]).tolist() == [True]
    assert api.nms_cpp(np.zeros((0, 4), np.int32), np.zeros(0)).shape == (0,)


def test_nms_cpp_matches_oracle(built, model_file):
    """Dialect CPP NMS of libjda vs the oracle's restatement of the multimap loop."""
    from jda_amd import api
    rng = np.random.default_rng(1)
    for trial in range(20):
        n = int(rng.integers(1, 80))
        s = rng.integers(10, 40, n)
        rc = np.c_[rng.integers(0, 50, n), rng.integers(0, 50, n), s, s].astype(np.int32)
        sc = np.round(rng.normal(0, 1, n), 1)       # ties likely
        # Python restatement of cascador.cpp:387-429
        order = sorted(range(n), key=lambda i: sc[i])    # stable ascending == multimap order
        alive, picked = list(order), []
        while alive:
            last = alive[-1]
            picked.append(last)
            nxt = []
            for i in alive:
                x1 = max(rc[i, 0], rc[last, 0]); y1 = max(rc[i, 1], rc[last, 1])
                x2 = min(rc[i, 0] + rc[i, 2], rc[last, 0] + rc[last, 2]); y2 = min(rc[i, 1] + rc[i, 3], rc[last, 1] + rc[last, 3])
                w, h = max(0., float(x2 - x1)), max(0., float(y2 - y1))
                ov = w * h / (float(rc[i, 2] * rc[i, 3]) + float(rc[last, 2] * rc[last, 3]) - w * h)
                if not ov > 0.3:
                    nxt.append(i)
            alive = nxt
        assert api.nms_cpp(rc, sc, 0.3).tolist() == picked, trial
