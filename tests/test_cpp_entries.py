"""Dialect CPP as a product (r06): jdaDetectBatchCppDevice (frames resident in HBM) and jdaDetectBatchCppRagged[Device]
(a list of differently sized images as ONE job -- literally the reference's `jda fddb` loop, one
joincascador.Detect(gray, ...) per image with fddb.method = 1, src/test.cpp:100-170 line 142,
src/jda/cascador.cpp:310-376,431-477).  Image by image the results must be those of jdaDetectBatchCpp on that image
alone and those of the oracle's restatement of Detect -- bit-exact (fp64).  PARITY UNPINNED like all of dialect CPP:
the oracle restates src/jda, nothing here can compile it (no OpenCV)."""
import numpy as np
import pytest

from conftest import same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda", 0)


def _eq(a, b, what=""):
    for k in ("rects", "scores", "shapes"):
        assert same(a[k], b[k]), (what, k, a[k].shape, b[k].shape)


def _images(sizes, seed=0):
    from jda_amd import synth
    return [synth.make_frames(1, w, h, seed=seed, first=i)[0] for i, (w, h) in enumerate(sizes)]


# sides below the minimum window (no level), exactly the minimum, odd sizes, tall and wide strips, sizes that end a level
SIZES = [(200, 150), (131, 97), (64, 48), (333, 250), (47, 200), (20, 20), (19, 60), (24, 23), (257, 255), (400, 123),
         (123, 400), (160, 160), (161, 159), (450, 450), (21, 20), (90, 300)]


def test_cpp_ragged_equals_per_image_calls_and_the_oracle(built, gpu, model_file):
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-1.0, norm_every=5)
    imgs = _images(SIZES, seed=11)
    c, o = api.Cascador(p), Oracle(p)
    got, st = c.detect_ragged_cpp(imgs, stats=True)
    assert len(got) == len(imgs)
    n_det = 0
    for i, im in enumerate(imgs):
        _eq(got[i], c.detect_batch_cpp(im[None])[0], i)        # the per-image entry, as fddb() calls Detect
        _eq(got[i], o.detect_cpp(im), i)                       # cascador.cpp:310-376,431-477 restated
        n_det += len(got[i]["scores"])
    assert n_det > 0
    assert st["patch_n"] == sum(synth.count_windows_cpp(w, h, 20, 5, 1.2) for w, h in SIZES)
    # the job's counters are the sums of the per-image calls' (Validate's n, cascador.cpp:187)
    per = [c.detect_batch_cpp(im[None], stats=True)[1] for im in imgs]
    assert st["cart_total_n"] == sum(s["cart_total_n"] for s in per)
    assert st["face_patch_n"] == sum(s["face_patch_n"] for s in per)
    # other call parameters, NMS off (every survivor, scan order)
    kw = dict(minimum_size=30, step=7, factor=1.5, overlap=0.5)
    got = c.detect_ragged_cpp(imgs, nms=False, **kw)
    for i, im in enumerate(imgs):
        _eq(got[i], c.detect_batch_cpp(im[None], nms=False, **kw)[0], i)
        _eq(got[i], o.detect_cpp(im, 30, 7, 1.5, 0.5, False), i)
    # a second job over a permutation of the list: results follow their images
    perm = [5, 0, 13, 2, 9, 7, 1]
    got2 = c.detect_ragged_cpp([imgs[j] for j in perm], **kw)
    want = c.detect_ragged_cpp(imgs, **kw)
    for k, j in enumerate(perm):
        _eq(got2[k], want[j], (k, j))


def test_cpp_ragged_edge_cases(built, gpu, model_file):
    from jda_amd import api
    p, _ = model_file((2, 8, 5, 3), 8, seed=5, cart_th=-0.5)
    c = api.Cascador(p)
    assert c.detect_ragged_cpp([]) == []
    tiny = _images([(19, 19), (30, 19), (19, 500)])                      # no image holds a 20-pixel window
    for r in c.detect_ragged_cpp(tiny):
        assert len(r["scores"]) == 0 and r["rects"].shape == (0, 4) and r["shapes"].shape == (0, 10)
    one = _images([(20, 20)], seed=2)                                    # exactly one candidate window
    _eq(c.detect_ragged_cpp(one)[0], c.detect_batch_cpp(one[0][None])[0])
    same_size = _images([(120, 90)] * 5, seed=3)                         # a uniform batch through the ragged entry
    got = c.detect_ragged_cpp(same_size)
    want = c.detect_batch_cpp(np.stack(same_size))
    for a, b in zip(got, want):
        _eq(a, b)
    with pytest.raises(api.JdaError):
        c.detect_ragged_cpp_packed(np.zeros(10, np.uint8), [0], [0], [5])    # an image without pixels
    with pytest.raises(api.JdaError):
        c.detect_ragged_cpp(same_size, factor=1.0)                       # the reference's loop would not terminate


def test_cpp_ragged_in_several_chunks_and_from_device_memory(built, gpu, model_file):
    """More chunks than lanes (ragged_chunk_windows_cpp forced small): the software pipeline reuses lanes; packed host
    buffer, separate host allocations and device-resident images agree."""
    import torch
    from jda_amd import api, synth
    p, _ = model_file((3, 70, 9, 5), 8, seed=71, cart_th=-0.9, norm_every=9)
    rng = np.random.default_rng(4)
    sizes = [(int(rng.integers(20, 260)), int(rng.integers(20, 200))) for _ in range(41)]
    imgs = _images(sizes, seed=9)
    c = api.Cascador(p)
    want = [c.detect_batch_cpp(im[None])[0] for im in imgs]
    assert sum(len(w["scores"]) for w in want) > 0
    offs, tot = [], 0
    for im in imgs:
        offs.append(tot); tot += im.size
    buf = np.concatenate([im.reshape(-1) for im in imgs])
    ws, hs = [s[0] for s in sizes], [s[1] for s in sizes]
    for chunk in (6000000, 200000, 20000):
        c.set_option("ragged_chunk_windows_cpp", chunk)
        got_list, st = c.detect_ragged_cpp(imgs, stats=True)
        got_buf = c.detect_ragged_cpp_packed(buf, offs, ws, hs)
        got_dev = c.detect_ragged_cpp_packed(torch.from_numpy(buf).to(gpu), offs, ws, hs)
        for i in range(len(imgs)):
            _eq(got_list[i], want[i], (chunk, i)); _eq(got_buf[i], want[i], (chunk, i)); _eq(got_dev[i], want[i], (chunk, i))
        assert st["patch_n"] == sum(synth.count_windows_cpp(w, h, 20, 5, 1.2) for w, h in sizes)


def test_cpp_device_entry_equals_the_host_entry(built, gpu, model_file):
    """jdaDetectBatchCppDevice (frames resident, padded stride) == jdaDetectBatchCpp == oracle."""
    import ctypes as C
    import torch
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-1.0, norm_every=5)
    frames = synth.make_frames(5, 210, 140, seed=21)
    c, o = api.Cascador(p), Oracle(p)
    want = c.detect_batch_cpp(frames)
    got, st = c.detect_batch_cpp_device(torch.from_numpy(frames).to(gpu), stats=True)
    for i in range(len(frames)):
        _eq(got[i], want[i], i)
        _eq(got[i], o.detect_cpp(frames[i]), i)
    assert st["patch_n"] == 5 * synth.count_windows_cpp(210, 140, 20, 5, 1.2)
    # a padded frame stride and a misaligned base
    stride = 210 * 140 + 77
    buf = torch.zeros(3 + 5 * stride, dtype=torch.uint8, device=gpu)
    for i in range(5):
        buf[3 + i * stride: 3 + i * stride + 210 * 140] = torch.from_numpy(frames[i].reshape(-1)).to(gpu)
    res = (api.jdaResultD * 5)()
    rc = api.lib.jdaDetectBatchCppDevice(c.h, C.c_void_p(buf.data_ptr() + 3), stride, 5, 210, 140, 20, 5, 1.2, 0.3, 1, None, res)
    assert rc == 0, api.last_error()
    for i in range(5):
        _eq(api._take_d(res[i]), want[i], i)
    # a stride smaller than a frame is refused
    rc = api.lib.jdaDetectBatchCppDevice(c.h, C.c_void_p(buf.data_ptr()), 100, 5, 210, 140, 20, 5, 1.2, 0.3, 1, None, res)
    assert rc == -1 and "frame_stride" in api.last_error()


def test_cpp_ragged_multi_scale_model_runs_image_by_image(built, gpu, model_file):
    """Split nodes that read the half / quarter image: the ragged scan does not cover them, the job falls back to one
    pass per image inside the entry -- same results."""
    from jda_amd import api
    from oracle.pyoracle import Oracle
    p, _ = model_file((2, 8, 5, 3), 8, seed=9, cart_th=-0.5, multi_scale=True)
    imgs = _images([(120, 90), (64, 80), (33, 47)], seed=4)
    c, o = api.Cascador(p), Oracle(p)
    got = c.detect_ragged_cpp(imgs)
    for i, im in enumerate(imgs):
        _eq(got[i], c.detect_batch_cpp(im[None])[0], i)
        _eq(got[i], o.detect_cpp(im), i)


def test_cpp_ragged_fddb_shaped_job_on_the_shipped_dimensions(built, gpu):
    """A slice of the FDDB-shaped job (BASELINE configs[3] sizes) on the 5 x 540-cart 27-landmark model through the
    dialect the reference's fddb() runs: ragged job == per-image calls, and the oracle on sampled images."""
    import os
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    path = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
    if not os.path.exists(path):
        m = synth.make_model(5, 540, 27, 4, seed=1)
        synth.calibrate_thresholds(m, synth.make_frames(8, 640, 480, seed=0, first=10_000_000))
        m.save(path + ".tmp", 8); os.replace(path + ".tmp", path)
    rng = np.random.default_rng(6)
    sizes = []
    for _ in range(96):
        long_side, short = int(rng.integers(300, 451)), int(rng.integers(225, 340))
        sizes.append((long_side, short) if rng.random() < 0.7 else (short, long_side))
    imgs = _images(sizes, seed=31)
    c = api.Cascador(path)
    got, st = c.detect_ragged_cpp(imgs, stats=True)
    assert st["patch_n"] == sum(synth.count_windows_cpp(w, h, 20, 5, 1.2) for w, h in sizes)
    for i in range(0, len(imgs), 7):
        _eq(got[i], c.detect_batch_cpp(imgs[i][None])[0], i)
    o = Oracle(path)
    for i in (0, 41, 95):
        _eq(got[i], o.detect_cpp(imgs[i]), i)
    assert sum(len(g["scores"]) for g in got) > 0


def _pack_d(results, L, frame_offset=0):
    """What jdaResultsDPack makes of per-image results: rows [frame, x, y, w, h, score, shape...] of doubles."""
    rows = []
    for i, r in enumerate(results):
        n = len(r["scores"])
        if n:
            m = np.empty((n, 6 + 2 * L), np.float64)
            m[:, 0] = frame_offset + i; m[:, 1:5] = r["rects"]; m[:, 5] = r["scores"]; m[:, 6:] = r["shapes"]
            rows.append(m)
    return np.concatenate(rows) if rows else np.empty((0, 6 + 2 * L), np.float64)


def test_cpp_ragged_rows_entries_equal_the_packed_results(built, gpu, model_file):
    """jdaDetectBatchCppRagged[Device]Rows: the job's rows, written in place by the library (per-image NMS in parallel,
    rows of an image at its prefix offset), must be bit for bit what the per-image jdaResultDs pack to -- several
    chunks, NMS on and off, a frame offset, host and device images, an empty job, images below the minimum window, and
    a multi-scale model (per-image fallback inside)."""
    import torch
    from jda_amd import api
    p, _ = model_file((3, 70, 9, 5), 8, seed=71, cart_th=-0.9, norm_every=9)
    rng = np.random.default_rng(14)
    sizes = [(int(rng.integers(15, 260)), int(rng.integers(15, 200))) for _ in range(43)]
    imgs = _images(sizes, seed=21)
    offs, tot = [], 0
    for im in imgs:
        offs.append(tot); tot += im.size
    buf = np.concatenate([im.reshape(-1) for im in imgs])
    d_buf = torch.from_numpy(buf).to(gpu)
    ws, hs = np.array([s[0] for s in sizes], np.int32), np.array([s[1] for s in sizes], np.int32)
    offs = np.array(offs, np.uint64)
    c = api.Cascador(p)
    assert hasattr(api.lib, "jdaDetectBatchCppRaggedDeviceRows")
    n_rows = 0
    for chunk in (6000000, 50000):
        c.set_option("ragged_chunk_windows_cpp", chunk)
        for nms in (True, False):
            want = _pack_d(c.detect_ragged_cpp_packed(buf, offs, ws, hs, nms=nms), c.L, frame_offset=11)
            for src in (buf, d_buf):
                rows = c.detect_ragged_cpp_packed(src, offs, ws, hs, nms=nms, keep_results="packed", frame_offset=11)
                assert rows.dtype == np.float64 and same(np.array(rows), want), (chunk, nms, rows.shape, want.shape)
            n_rows = max(n_rows, len(want))
    assert n_rows > 0
    kept = c.detect_ragged_cpp_packed(d_buf, offs, ws, hs, keep_results="packed")      # the array owns the library's buffer
    view = kept[len(kept) // 2:]
    ref = np.array(kept, copy=True)
    del kept
    assert same(np.array(view), ref[len(ref) // 2:])
    assert c.detect_ragged_cpp_packed(d_buf, offs[:0], ws[:0], hs[:0], keep_results="packed").shape == (0, 6 + 2 * c.L)
    tiny = _images([(19, 60), (12, 12)])
    tb = np.concatenate([t.reshape(-1) for t in tiny])
    assert c.detect_ragged_cpp_packed(tb, [0, 19 * 60], [19, 12], [60, 12], keep_results="packed").shape == (0, 6 + 2 * c.L)
    pm, _ = model_file((2, 8, 5, 3), 8, seed=9, cart_th=-0.5, multi_scale=True)
    cm = api.Cascador(pm)
    want = _pack_d(cm.detect_ragged_cpp_packed(buf, offs, ws, hs), cm.L, frame_offset=3)
    assert same(np.array(cm.detect_ragged_cpp_packed(buf, offs, ws, hs, keep_results="packed", frame_offset=3)), want)


@pytest.mark.parametrize("sim", [False, True])
@pytest.mark.parametrize("hdr", [(1, 6), (0, 11), (2, -1), (0, -1), (2, 19)])
def test_cpp_on_a_model_still_in_training_stops_where_validate_stops(built, gpu, tmp_path, hdr, sim):
    """A trainer snapshot (jda_xxxx_stage_s_cart_c.model) carries its training status in the header, and the reference's
    Validate honours it: stages [0, s) in full, then carts [0, c] of stage s WITHOUT that stage's regression
    (cascador.cpp:177-209, 84-104).  Found by the second reading of src/jda (oracle/cpp_reading2.py,
    tests/test_cpp_second_reading.py): until then both the oracle and the kernels ran all T x K carts in dialect CPP, as
    dialect C's reference does (c/jda.c:499-505 drops the two ints).  The oracle runs Validate's literal loop bounds; the
    product pads its fp64 tables with pass-through carts and zero weight rows (model_dev.cpp) -- window by window the reject
    length, score, leaf path and shape must be the same, and so must every entry's detections.  Dialect C on the same
    file keeps running everything.  sim: with the similarity transform on, the stage in training walks with the parameter the
    stage before it computed -- Validate does not recompute stp_mc for it (cascador.cpp:178-200) -- or, being the first stage,
    with STParameter's default; both restatements read it that way (tests/test_cpp_second_reading.py)."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    dims = (3, 20, 5, 4)
    mdl = synth.make_model(*dims, seed=3, cart_th=-1.0, norm_every=5, w_sigma=2e-2 if sim else 2e-3)
    p = str(tmp_path / "snapshot.model")
    mdl.save(p, 8, header_stage=hdr[0], header_cart=hdr[1])
    full = str(tmp_path / "full.model")
    mdl.save(full, 8)
    imgs = _images([(131, 97), (64, 48), (200, 150)], seed=5)
    c, o = api.Cascador(p), Oracle(p)
    c.set_similarity_transform(sim); o.set_similarity_transform(sim)
    ran = hdr[0] * dims[1] + hdr[1] + 1                       # carts Validate runs for a window that passes them all
    for im in imgs[:2]:
        got, want = c.trace_cpp(im[None]), o.trace_cpp(im)
        faces = got["carts_n"] == dims[0] * dims[1]                 # (a face walks the padding too; nothing but this counter sees it)
        assert (want["carts_n"][faces] == ran).all()                # ... where Validate stopped after every cart it runs
        assert np.array_equal(got["carts_n"][~faces], want["carts_n"][~faces])      # rejected windows: Validate's n
        assert np.array_equal(got["path_hash"][~faces], want["path_hash"][~faces])  # ... and the same leaves on the way
        for k in ("score", "shapes"):
            assert same(got[k], want[k]), k
        every = o.detect_cpp(im, nms=False)                         # every face, scan order
        _eq(c.detect_batch_cpp(im[None], nms=False)[0], every, "nms off")
        assert 0 < len(every["scores"]) <= faces.sum() and (hdr == (0, -1) or not faces.all())   # (<: a window rejected by the LAST cart also counts T x K)
    got = c.detect_ragged_cpp(imgs)
    for i, im in enumerate(imgs):
        _eq(got[i], o.detect_cpp(im), i)
        _eq(got[i], c.detect_batch_cpp(im[None])[0], i)
    o.set_similarity_transform(False)
    # the same file through dialect C: all T x K carts, like the complete model (c/jda.c ignores the status)
    cf = api.Cascador(full)
    a, b = c.detect(imgs[2]), cf.detect(imgs[2])
    for k in ("bboxes", "scores", "shapes"):
        assert same(a[k], b[k]), k
    c.close(); cf.close(); o.close()
