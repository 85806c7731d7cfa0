"""jdaDetectBatchRagged: a list of differently sized images as ONE job (the reference's FDDB loop, one Detect per
image, src/test.cpp:100-170) must give, image by image, exactly what jdaDetect gives on that image -- and therefore
what the oracle / the compiled reference c/jda.c:443-480 give.  Bit-exact."""
import os

import numpy as np
import pytest

from conftest import S_DIMS, same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda", 0)


def _eq(a, b, what=""):
    for k in ("bboxes", "scores", "shapes"):
        assert same(a[k], b[k]), (what, k, a[k].shape, b[k].shape)


def _images(sizes, seed=0):
    from jda_amd import synth
    return [synth.make_frames(1, w, h, seed=seed, first=i)[0] for i, (w, h) in enumerate(sizes)]


SIZES = [(200, 150), (131, 97), (64, 48), (333, 250), (47, 200), (46, 46), (45, 60), (20, 20), (257, 255), (400, 123),
         (123, 400), (160, 160), (161, 159), (450, 450), (48, 47), (90, 300)]


def test_ragged_equals_per_image_calls_and_the_oracle(built, gpu, model_file):
    from jda_amd import api
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-1.0, norm_every=5)
    imgs = _images(SIZES, seed=11)
    c, o = api.Cascador(p), Oracle(p)
    got, st = c.detect_ragged(imgs, stats=True)
    assert len(got) == len(imgs)
    n_det = 0
    for i, im in enumerate(imgs):
        _eq(got[i], c.detect(im), i)                       # the drop-in call, one image at a time
        _eq(got[i], o.detect(im), i)                       # c/jda.c:443-480 restated
        n_det += len(got[i]["scores"])
    assert n_det > 0
    assert st["patch_n"] == sum(api.count_windows(w, h)[0] for w, h in SIZES)
    # other call parameters, NMS off (every survivor, scan order)
    kw = dict(scale=1.5, min_size=30, max_size=120, th=0.0)
    got = c.detect_ragged(imgs, nms=False, **kw)
    for i, im in enumerate(imgs):
        _eq(got[i], c.detect_batch(im[None], nms=False, **kw)[0], i)
    # a second job over a permutation of the list: results follow their images
    perm = [5, 0, 13, 2, 9, 7, 1]
    got2 = c.detect_ragged([imgs[j] for j in perm], **kw)
    want = c.detect_ragged(imgs, **kw)
    for k, j in enumerate(perm):
        _eq(got2[k], want[j], (k, j))


def test_ragged_edge_cases(built, gpu, model_file):
    from jda_amd import api
    p, _ = model_file((2, 8, 5, 3), 8, seed=5, cart_th=-0.5)
    c = api.Cascador(p)
    assert c.detect_ragged([]) == []
    tiny = _images([(20, 20), (30, 39), (39, 500)])                      # no image holds a 46-pixel window
    for r in c.detect_ragged(tiny):
        assert len(r["scores"]) == 0 and r["bboxes"].shape == (0, 3) and r["shapes"].shape == (0, 10)
    one = _images([(46, 46)], seed=2)                                    # exactly one candidate window
    _eq(c.detect_ragged(one)[0], c.detect(one[0]))
    same_size = _images([(120, 90)] * 5, seed=3)                         # a uniform batch through the ragged entry
    got = c.detect_ragged(same_size)
    want = c.detect_batch(np.stack(same_size))
    for a, b in zip(got, want):
        _eq(a, b)
    with pytest.raises(api.JdaError):
        c.detect_ragged_packed(np.zeros(10, np.uint8), [0], [0], [5])    # an image without pixels


def test_ragged_in_several_chunks_and_from_device_memory(built, gpu, model_file):
    """More chunks than lanes (ragged_chunk_windows forced small): the software pipeline reuses lanes; packed host
    buffer (one H2D copy), separate host allocations (pinned staging) and device-resident images agree."""
    import torch
    from jda_amd import api
    p, _ = model_file((3, 70, 9, 5), 8, seed=71, cart_th=-0.9, norm_every=9)
    rng = np.random.default_rng(4)
    sizes = [(int(rng.integers(46, 260)), int(rng.integers(46, 200))) for _ in range(41)]
    imgs = _images(sizes, seed=9)
    c = api.Cascador(p)
    want = [c.detect(im) for im in imgs]
    offs, tot = [], 0
    for im in imgs:
        offs.append(tot); tot += im.size
    buf = np.concatenate([im.reshape(-1) for im in imgs])
    ws, hs = [s[0] for s in sizes], [s[1] for s in sizes]
    for chunk in (6000000, 60000, 7000):
        c.set_option("ragged_chunk_windows", chunk)
        got_list, st = c.detect_ragged(imgs, stats=True)
        got_buf = c.detect_ragged_packed(buf, offs, ws, hs)
        got_dev = c.detect_ragged_packed(torch.from_numpy(buf).to(gpu), offs, ws, hs)
        for i in range(len(imgs)):
            _eq(got_list[i], want[i], (chunk, i)); _eq(got_buf[i], want[i], (chunk, i)); _eq(got_dev[i], want[i], (chunk, i))
        assert st["patch_n"] == sum(api.count_windows(w, h)[0] for w, h in sizes)
    # the counters of the job are the sums of the per-image calls
    tot_c = sum(c.detect_batch(im[None], stats=True)[1]["cart_total_n"] for im in imgs)
    assert st["cart_total_n"] == tot_c


def test_ragged_models_outside_the_fast_scan_run_image_by_image(built, gpu, model_file):
    """Multi-scale split nodes (half/quarter images) and all-pass cascades (dense kernel) are served by per-image
    passes inside the ragged entry: same results."""
    from jda_amd import api, synth
    imgs = _images([(150, 110), (97, 131), (200, 64), (300, 200), (222, 333), (280, 280)], seed=21)
    m = synth.make_model(2, 8, 5, 3, seed=7, cart_th=-0.6, multi_scale=True)
    import tempfile
    d = tempfile.mkdtemp()
    p = os.path.join(d, "ms.model"); m.save(p, 8)
    c = api.Cascador(p)
    for a, im in zip(c.detect_ragged(imgs), imgs):
        _eq(a, c.detect(im))
    p2, _ = model_file((3, 20, 5, 4), 8, seed=63)                        # all-pass: every window survives every cart
    c2 = api.Cascador(p2)
    want = [c2.detect(im, th=0.2) for im in imgs]
    c3 = api.Cascador(p2)
    dense = []
    for rep in range(2):                  # the first job finds that nothing is rejected; the second runs dense, per image
        got, st = c3.detect_ragged(imgs, th=0.2, stats=True)
        dense.append(st["dense_passes"])
        for a, b in zip(got, want):
            _eq(a, b)
    assert dense[0] == 0 and dense[1] > 0, dense


def test_ragged_fddb_sized_job_shipped_dims(built, gpu, tmp_path):
    """BASELINE.json configs[3]-shaped: 300 FDDB-sized images (<= 450x450, varied aspect), shipped model dimensions,
    cascade regime, as one ragged job vs per-image jdaDetect (all) and the oracle + compiled reference (sampled)."""
    import json
    import time
    from jda_amd import api, synth
    from oracle import pyoracle
    rng = np.random.default_rng(0)
    sizes = []
    for _ in range(300):
        long_side = int(rng.integers(300, 451)); short = int(rng.integers(225, long_side + 1))
        sizes.append((long_side, short) if rng.random() < 0.5 else (short, long_side))
    imgs = _images(sizes, seed=1)
    m = synth.make_model(*S_DIMS, seed=1)
    synth.calibrate_thresholds(m, synth.make_frames(8, 450, 450, seed=0, first=10_000_000), scale=1.25, min_size=40)
    p = str(tmp_path / "fddb.model"); m.save(p, 8)
    c = api.Cascador(p)
    got, st = c.detect_ragged(imgs, stats=True)                          # (first job: plans, tables, workspace)
    t0 = time.perf_counter()
    got2, st2 = c.detect_ragged(imgs, stats=True)
    el = time.perf_counter() - t0
    assert st["patch_n"] == sum(api.count_windows(w, h)[0] for w, h in sizes)
    for k in ("cart_total_n", "face_patch_n", "cart_gothrough_n"):
        assert st[k] == st2[k], k
    for i, im in enumerate(imgs):
        _eq(got[i], got2[i], i)
        _eq(got[i], c.detect(im), i)
    o = pyoracle.Oracle(p)
    ref = pyoracle.Reference(p, S_DIMS, 8) if pyoracle.reference_lib_path(*S_DIMS) else None
    for i in (0, 77, 299):
        _eq(got[i], o.detect(imgs[i]), i)
        if ref is not None:
            _eq(got[i], ref.detect(imgs[i]), i)
    assert sum(len(g["scores"]) for g in got) > 0
    print("ragged 300 FDDB-sized images: %.2f ms, %.0f images/s, %.3e windows/s" % (el * 1e3, 300 / el, st["patch_n"] / el))


@pytest.mark.parametrize("dims,th", [((3, 20, 5, 4), -1.0), (S_DIMS, -2.0)])
def test_ragged_chunks_through_the_persistent_scan(built, gpu, model_file, monkeypatch, dims, th):
    """The single-level launches of a ragged chunk go through k_scan_p's RAGGED instantiation (tiles named by the chunk's
    block map, the level's tile re-cut per image, the image's own window grid travelling with the slot): same results,
    bit for bit, as with the closed-tile k_scan and as jdaDetect image by image -- 300 images of mixed sizes (some with
    fewer levels than others, one too small for any window) in several chunks, scan_p = 2 so that every level it fits
    takes the persistent form."""
    from jda_amd import api, synth
    p, _ = model_file(dims, 8, seed=3, cart_th=th, norm_every=5)
    rng = np.random.default_rng(9)
    base = synth.make_frames(8, 360, 300, seed=12)
    imgs = []
    for i in range(300):
        w, h = int(rng.integers(60, 361)), int(rng.integers(50, 301))
        imgs.append(np.ascontiguousarray(base[i % 8][:h, :w]))
    imgs[17] = np.ascontiguousarray(base[0][:20, :30])
    monkeypatch.setenv("JDA_SCAN_P", "2")
    monkeypatch.setenv("JDA_RAGGED_CHUNK_WINDOWS", "700000"); monkeypatch.setenv("JDA_RAGGED_CHUNK_MIN_WINDOWS", "100000")
    on = api.Cascador(p)
    monkeypatch.setenv("JDA_SCAN_P_RAGGED", "0")
    off = api.Cascador(p)
    assert on.get_option("scan_p_ragged") == 1 and off.get_option("scan_p_ragged") == 0
    (a, sa), (b, sb) = on.detect_ragged(imgs, stats=True), off.detect_ragged(imgs, stats=True)
    assert len(a) == len(b) == 300
    for i in range(300):
        _eq(a[i], b[i], i)
    for k in ("patch_n", "face_patch_n", "cart_gothrough_n", "cart_total_n", "handoff_n", "scan_cart_n", "scan_patch_n"):
        assert sa[k] == sb[k], k
    assert sa["scan_patch_n"] == sa["patch_n"] == sum(api.count_windows(im.shape[1], im.shape[0])[0] for im in imgs)
    for i in (0, 17, 123, 299):
        _eq(a[i], on.detect(imgs[i]), ("jdaDetect", i))
    assert sum(len(r["scores"]) for r in a) > 0 and len(a[17]["scores"]) == 0
    # second job (the finishing launches are sized by prediction now), NMS off
    for x, y in zip(on.detect_ragged(imgs, nms=False), off.detect_ragged(imgs, nms=False)):
        _eq(x, y, "second job")


def test_ragged_rows_entries_equal_the_packed_results(built, gpu, model_file, monkeypatch):
    """jdaDetectBatchRagged[Device]Rows: the job's detections as one matrix of rows must be, bit for bit, what
    jdaResultsPack makes of the n jdaResults of the same job -- device-post-processed chunks (k_post), chunks too
    small for it (host NMS), several chunks, NMS off, a frame offset, an empty job, images without a window."""
    import torch
    from jda_amd import api, dist as jd
    p, _ = model_file((3, 70, 9, 5), 8, seed=71, cart_th=-0.9, norm_every=9)
    rng = np.random.default_rng(8)
    sizes = [(int(rng.integers(30, 260)), int(rng.integers(30, 200))) for _ in range(45)]
    imgs = _images(sizes, seed=13)
    offs, tot = [], 0
    for im in imgs:
        offs.append(tot); tot += im.size
    buf = np.concatenate([im.reshape(-1) for im in imgs])
    ws, hs = np.array([s[0] for s in sizes], np.int32), np.array([s[1] for s in sizes], np.int32)
    offs = np.array(offs, np.uint64)
    d_buf = torch.from_numpy(buf).to(gpu)
    c = api.Cascador(p)
    assert hasattr(api.lib, "jdaDetectBatchRaggedDeviceRows")
    for chunk in (6000000, 30000):
        c.set_option("ragged_chunk_windows", chunk)
        for nms in (True, False):
            res = c.detect_ragged_packed(buf, offs, ws, hs, nms=nms)                       # n jdaResults
            want = jd.pack_detections(res, c.L, frame_offset=100)
            for src in (buf, d_buf):
                rows, st = c.detect_ragged_packed(src, offs, ws, hs, nms=nms, keep_results="packed", frame_offset=100, stats=True)
                assert rows.dtype == np.float32 and same(rows, want), (chunk, nms, rows.shape, want.shape)
                assert st["patch_n"] == sum(api.count_windows(w, h)[0] for w, h in sizes)
    assert len(want) > 0
    # few images: below device_post_min_frames the host post-processes; none: an empty matrix
    few = c.detect_ragged_packed(d_buf, offs[:5], ws[:5], hs[:5], keep_results="packed")
    assert same(few, jd.pack_detections(c.detect_ragged_packed(d_buf, offs[:5], ws[:5], hs[:5]), c.L))
    none = c.detect_ragged_packed(d_buf, offs[:0], ws[:0], hs[:0], keep_results="packed")
    assert none.shape == (0, 5 + 2 * c.L)
    tiny = _images([(20, 20), (30, 39)])
    tb = np.concatenate([t.reshape(-1) for t in tiny])
    assert c.detect_ragged_packed(tb, [0, 400], [20, 30], [20, 39], keep_results="packed").shape == (0, 5 + 2 * c.L)
    # a model the ragged scan does not cover (multi-scale split nodes) runs image by image inside: same rows
    pm, _ = model_file((2, 12, 5, 3), 8, seed=9, cart_th=-0.8, multi_scale=True)
    cm = api.Cascador(pm)
    res = cm.detect_ragged_packed(buf, offs, ws, hs)
    assert same(cm.detect_ragged_packed(buf, offs, ws, hs, keep_results="packed", frame_offset=7), jd.pack_detections(res, cm.L, frame_offset=7))
