"""k_post, the per-frame post-processing of a dialect-C batch on the device (jda_amd/csrc/k_post.hip: detections back into
scan order, the reference's score order, greedy NMS, scan-order output, relocation; c/jda.c:237-316), must give exactly
what the host form gives (post.cpp, itself checked against the compiled reference: tests/test_nms.py, golden `score_ties`)
-- boxes, score bits, landmark bits, frame by frame -- including where it declines (a frame with more than 1,024
detections, ties among more than 256) and the host takes the pass.  `device_post` is read at every call."""
import numpy as np
import pytest

from conftest import S_DIMS, same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda", 0)


def _both(casc, dev, **kw):
    casc.set_option("device_post", 0)
    host = [casc.detect_batch_device(dev, **kw) for _ in range(2)][-1]
    casc.set_option("device_post", 1)
    outs = [casc.detect_batch_device(dev, **kw) for _ in range(3)]      # (the first pass on a plan has no prediction: host form)
    for got in outs:
        assert len(got) == len(host)
        for i, (a, b) in enumerate(zip(host, got)):
            for k in ("bboxes", "scores", "shapes"):
                assert same(a[k], b[k]), (i, k, len(a["scores"]), len(b["scores"]))
    return host


@pytest.mark.parametrize("nms", [True, False])
def test_device_post_equals_host_post_shipped_dimensions(built, gpu, model_file, nms):
    import torch
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file(S_DIMS, 8, seed=3, cart_th=-2.0, norm_every=5)
    frames = synth.make_frames(40, 320, 240, seed=21)
    c = api.Cascador(p)
    host = _both(c, torch.from_numpy(frames).cuda(), th=-0.5, nms=nms)
    assert sum(len(h["scores"]) for h in host) > 40
    o = Oracle(p)
    for i in ((0, 17, 39) if nms else ()):       # (the oracle's raw list is checked against the host form elsewhere)
        want = o.detect(frames[i], th=-0.5, nms=nms)
        for k in ("bboxes", "scores", "shapes"):
            assert same(want[k], host[i][k]), (i, k)
    c.close()


@pytest.mark.parametrize("dims,size,n", [((1, 4, 3, 2), (64, 64), 24),      # a handful of leaf values: ties everywhere, few windows (literal order on the device)
                                         ((1, 4, 3, 2), (200, 150), 20),   # ... and thousands of tied detections per frame (declined: host form)
                                         ((2, 8, 5, 3), (160, 120), 32)])
def test_ties_and_crowded_frames(built, gpu, model_file, dims, size, n):
    import torch
    from jda_amd import api, synth
    p, _ = model_file(dims, 8, seed=5, cart_th=synth.NEG_BIG)      # nothing is rejected: every window is a detection
    frames = synth.make_frames(n, size[0], size[1], seed=22)
    c = api.Cascador(p)
    host = _both(c, torch.from_numpy(frames).cuda(), th=-3.0e38)
    assert sum(len(h["scores"]) for h in host) > 0
    c.close()


def test_two_lanes_and_sub_batches(built, gpu, model_file, monkeypatch):
    """A batch that goes through two lanes in several sub-batches (workspace of 64 MB): every pass posts its own frames."""
    import torch
    from jda_amd import api, synth
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-1.0, norm_every=5)
    frames = synth.make_frames(96, 200, 150, seed=23)
    monkeypatch.setenv("JDA_WORKSPACE_MB", "8"); monkeypatch.setenv("JDA_LANES_MIN_WINDOWS", "1000")
    c = api.Cascador(p)
    host = _both(c, torch.from_numpy(frames).cuda(), th=-0.5)
    assert sum(len(h["scores"]) for h in host) > 96
    c.close()


def test_submit_wait_tickets_use_it_too(built, gpu, model_file):
    import torch
    from jda_amd import api, synth
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-1.0, norm_every=5)
    frames = synth.make_frames(48, 200, 150, seed=24)
    dev = torch.from_numpy(frames).cuda()
    c = api.Cascador(p)
    c.set_option("device_post", 0)
    want = [c.detect_batch_device(dev, th=-0.5) for _ in range(2)][-1]
    c.set_option("device_post", 1)
    q = [c.submit_batch_device(dev, th=-0.5) for _ in range(2)]
    for k in range(5):
        if k < 3:
            q.append(c.submit_batch_device(dev, th=-0.5))
        got = c.wait_batch(q.pop(0))
        assert len(got) == len(want)
        for a, b in zip(want, got):
            for key in ("bboxes", "scores", "shapes"):
                assert same(a[key], b[key]), (k, key)
    assert sum(len(w["scores"]) for w in want) > 48
    c.close()


def test_ragged_chunks_posted_and_declined(built, gpu, model_file):
    """The ragged branch of k_post (an image's gids are one range of the chunk, its window grids are re-derived on the
    device from its size) against the host form, bit for bit: images of mixed sizes in several chunks, one too small
    for any window (n_lv == 0), a flat image whose 1,998 windows all pass with ONE score (crowded and tied: its chunk
    is declined and goes through the host form while the other chunks are posted), a small flat image (ties replayed
    literally on the device), from host memory and from one packed device buffer."""
    import torch
    from jda_amd import api, synth
    p, _ = model_file((3, 20, 5, 4), 8, seed=2, cart_th=-0.5, norm_every=5)
    base = synth.make_frames(8, 200, 150, seed=22)
    rng = np.random.default_rng(4)
    imgs = []
    for i in range(64):
        w, h = int(rng.integers(120, 201)), int(rng.integers(100, 151))
        imgs.append(np.ascontiguousarray(base[i % 8][:h, :w]))
    imgs[5] = np.ascontiguousarray(base[0][:20, :20])                 # no window fits
    imgs[9] = np.full((150, 200), 128, np.uint8)                      # every window passes, every score equal
    imgs[44] = np.full((60, 70), 90, np.uint8)                        # a handful of tied detections
    c = api.Cascador(p)
    c.set_option("ragged_chunk_windows", 25000); c.set_option("ragged_chunk_min_windows", 1000)
    c.set_option("device_post_min_frames", 4)
    offs, tot = [], 0
    for im in imgs:
        offs.append(tot); tot += im.size
    buf = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).cuda()
    ws, hs = [im.shape[1] for im in imgs], [im.shape[0] for im in imgs]
    for nms in (True, False):
        c.set_option("device_post", 0)
        host = [c.detect_ragged(imgs, th=-0.5, nms=nms) for _ in range(2)][-1]
        c.set_option("device_post", 1)
        for rep in range(3):                                          # (the first job on a plan has no prediction: host form)
            for got in (c.detect_ragged(imgs, th=-0.5, nms=nms), c.detect_ragged_packed(buf, offs, ws, hs, th=-0.5, nms=nms)):
                assert len(got) == len(host) == 64
                for i, (a, b) in enumerate(zip(host, got)):
                    for k in ("bboxes", "scores", "shapes"):
                        assert same(a[k], b[k]), (nms, rep, i, k, len(a["scores"]), len(b["scores"]))
        assert len(host[5]["scores"]) == 0
        if not nms:
            assert len(host[9]["scores"]) == 1998 and len(np.unique(host[9]["scores"])) == 1
            assert 0 < len(host[44]["scores"]) <= 256 and len(np.unique(host[44]["scores"])) == 1
    # per image the ragged job is jdaDetect on that image
    full = c.detect_ragged(imgs, th=-0.5)
    for i in (0, 9, 44, 63):
        one = c.detect(imgs[i], th=-0.5)
        for k in ("bboxes", "scores", "shapes"):
            assert same(one[k], full[i][k]), (i, k)
    c.close()
