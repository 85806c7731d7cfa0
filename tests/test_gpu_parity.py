"""Parity of the HIP detect path (through the C ABI of libjda.so) against
 (1) the golden vectors the reference itself produced (tests/golden/),
 (2) the CPU oracle, window by window: carts evaluated (reject decision), score
     bits, leaf-path hash (tree paths / leaf indices), shape bits,
 (3) the reference's own compiled c/jda.c when its prebuilt library travelled,
 (4) size-independent properties at BASELINE.json's full sizes.
Bit-exact everywhere (the landmark tolerance north_star allows, 1e-5, is not used)."""
import os

import numpy as np
import pytest

import golden_util
from conftest import TINY_DIMS, S_DIMS, same, bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda", 0)


def _compare_trace(casc, orc, frames, **kw):
    g = casc.trace(frames, **kw)
    off = 0
    for i in range(len(frames)):
        r = orc.trace(frames[i], **kw)
        n = len(r["carts_n"])
        for k in ("carts_n", "score", "path_hash", "shapes"):
            assert same(r[k], g[k][off:off + n]), (i, k, int((bits(r[k]) != bits(g[k][off:off + n])).sum()))
        off += n
    assert off == len(g["carts_n"])


def _compare_detect(got, want):
    for k in ("bboxes", "scores", "shapes"):
        assert same(got[k], want[k]), k


# ---------------------------------------------------------------- golden vectors

@pytest.mark.parametrize("name", golden_util.NAMES)
def test_golden_vectors(built, gpu, tmp_path, name):
    from jda_amd import api
    meta, g, mp = golden_util.load(name, tmp_path)
    scale, mn, mx, th = meta["call"]
    c = api.Cascador(mp, "double")
    post = c.detect(g["frame"], scale, 0.1, mn, mx, th)                       # the drop-in call
    raw = c.detect_batch(g["frame"][None], scale, mn, mx, th, nms=False)[0]
    for k in ("bboxes", "scores"):
        assert same(raw[k], g["raw_" + k]), (name, k)
        assert same(post[k], g["post_" + k]), (name, k)
    assert same(post["shapes"], g["post_shapes"]), name
    # pre-NMS shapes: ours are relocated; undo nothing -- relocate the reference's instead
    sz = g["raw_bboxes"][:, 2].astype(np.float32)[:, None]
    want = g["raw_shapes"].copy()
    want[:, 0::2] = want[:, 0::2] * sz + g["raw_bboxes"][:, 0].astype(np.float32)[:, None]
    want[:, 1::2] = want[:, 1::2] * sz + g["raw_bboxes"][:, 1].astype(np.float32)[:, None]
    assert same(raw["shapes"], want), name
    h, w = g["frame"].shape
    if min(w, h) >= 2:
        half, quarter = c.build_pyramid(g["frame"])
        assert np.array_equal(half, g["half"]) and np.array_equal(quarter, g["quarter"])


# ---------------------------------------------------------------- oracle, window by window

@pytest.mark.parametrize("dims", TINY_DIMS)
@pytest.mark.parametrize("th", [-3.0e38, -1.0, -0.3])
def test_trace_and_detect_vs_oracle(built, gpu, model_file, dims, th):
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file(dims, 8, seed=3, cart_th=th, norm_every=5)
    frames = synth.make_frames(2, 200, 150, seed=11)
    c, o = api.Cascador(p), Oracle(p)
    _compare_trace(c, o, frames)
    dets = c.detect_batch(frames, th=-0.5)
    for i in range(len(frames)):
        _compare_detect(dets[i], o.detect(frames[i], th=-0.5))


@pytest.mark.parametrize("w,h,scale,mn,mx", [(131, 97, 1.2, 30, -1), (320, 240, 1.4, 24, 100), (64, 64, 2.0, 30, 0),
                                             (450, 337, 1.25, 20, 333), (24, 24, 1.25, 0, -1), (23, 50, 1.25, 40, -1)])
def test_ragged_sizes_and_call_arguments(built, gpu, model_file, w, h, scale, mn, mx):
    """Odd widths (unaligned tile loads), max_size cuts, single-window and empty scans."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 20, 5, 4), 8, seed=5, cart_th=-0.8, norm_every=7)
    frames = synth.make_frames(3, w, h, seed=2)
    c, o = api.Cascador(p), Oracle(p)
    kw = dict(scale=scale, min_size=mn, max_size=mx)
    _compare_trace(c, o, frames, **kw)
    dets = c.detect_batch(frames, th=-0.5, **kw)
    for i in range(len(frames)):
        _compare_detect(dets[i], o.detect(frames[i], th=-0.5, **kw))


def test_multiscale_model_generic_walker(built, gpu, model_file):
    """scale != 0 nodes: pyramid built on device, guarded reads (documented divergence from the
    reference's out-of-bounds reads; identical to it when it is in bounds)."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 20, 5, 4), 8, seed=6, cart_th=-1.0, multi_scale=True)
    frames = synth.make_frames(2, 200, 150, seed=4)
    c, o = api.Cascador(p), Oracle(p)
    assert c.multi_scale
    _compare_trace(c, o, frames)
    for i, d in enumerate(c.detect_batch(frames)):
        _compare_detect(d, o.detect(frames[i]))
    half, quarter = c.build_pyramid(frames[0])
    hw, hh, qw, qh = o.pyramid_dims(200, 150)
    assert np.array_equal(half, o.resize(frames[0], hw, hh)) and np.array_equal(quarter, o.resize(frames[0], qw, qh))


def test_scan_and_generic_walker_agree(built, gpu, model_file, monkeypatch):
    """The LDS-tiled stage-0 scan and the generic walker are two implementations of stage 0."""
    from jda_amd import api, synth
    p, _ = model_file((3, 70, 9, 5), 8, seed=8, cart_th=-1.2, norm_every=9)
    frames = synth.make_frames(2, 320, 240, seed=6)
    a = api.Cascador(p).trace(frames)
    monkeypatch.setenv("JDA_NO_FAST_SCAN", "1")
    b = api.Cascador(p).trace(frames)
    for k in a:
        assert same(a[k], b[k]), k


def test_extreme_model_values(built, gpu, model_file, tmp_path):
    """Thresholds outside the byte-difference range, huge offsets (coordinates that overflow
    int32 in the reference's float->int cast), NaN/inf cart thresholds, denormal weights."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    m = synth.make_model(3, 20, 5, 4, seed=12, cart_th=-1.0, norm_every=4)
    m.nth[0, ::3, 0] = 100000; m.nth[1, ::4, 1] = -100000; m.nth[2, 1::2, 2] = 255; m.nth[0, 1::5, 1] = -256
    m.off[0, 2, 0, 0] = 3.0e9; m.off[1, 3, 1, 3] = -3.0e9; m.off[2, 0, 0, 1] = 1e30
    m.cth[1, 5] = float("-inf"); m.cth[2, 7] = float("nan")
    m.w[0, ::7] = 1e-41                       # denormal in f32
    m.leaf[0, 3, :] = 1e-42
    p = str(tmp_path / "extreme.model"); m.save(p, 8)
    frames = synth.make_frames(2, 160, 120, seed=9)
    c, o = api.Cascador(p), Oracle(p)
    _compare_trace(c, o, frames)
    for i, d in enumerate(c.detect_batch(frames, th=-5.0)):
        _compare_detect(d, o.detect(frames[i], th=-5.0))


# ---------------------------------------------------------------- shipped dimensions

def test_shipped_dims_vs_oracle_and_reference(built, gpu, model_file):
    from jda_amd import api, synth
    from oracle import pyoracle
    p, _ = model_file(S_DIMS, 8, seed=1, cart_th=-2.0)
    frames = synth.make_frames(2, 320, 240, seed=14)
    c, o = api.Cascador(p, "double"), pyoracle.Oracle(p)
    _compare_trace(c, o, frames)
    dets = c.detect_batch(frames)
    ref = pyoracle.Reference(p, S_DIMS, 8) if pyoracle.reference_lib_path(*S_DIMS) else None
    for i in range(len(frames)):
        _compare_detect(dets[i], o.detect(frames[i]))
        if ref is not None:
            _compare_detect(dets[i], ref.detect(frames[i]))
    assert sum(len(d["scores"]) for d in dets) > 0


def test_shipped_dims_allpass_small_frame(built, gpu, model_file):
    """Every window walks all 2,700 carts and gathers 5 x 540 weight rows."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file(S_DIMS, 4, seed=2)
    frames = synth.make_frames(1, 160, 120, seed=3)
    c, o = api.Cascador(p, "float"), Oracle(p)
    _compare_trace(c, o, frames)
    tr = c.trace(frames)
    assert (tr["carts_n"] == 2700).all()


# ---------------------------------------------------------------- full-size properties

def test_full_batch_properties(built, gpu, tmp_path):
    """BASELINE.json configs[1] at full size (256 frames 640x480, S dims, cascade regime):
    window accounting, run-to-run determinism, batch == single-frame calls == device-resident
    entry, counters consistent with an oracle-traced sample, reference agreement on a sample."""
    import torch
    from jda_amd import api, synth
    from oracle import pyoracle
    calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
    m = synth.make_model(*S_DIMS, seed=1)
    synth.calibrate_thresholds(m, calib)
    p = str(tmp_path / "cascade.model"); m.save(p, 8)
    frames = synth.make_frames(256, 640, 480, seed=0)
    c = api.Cascador(p)
    d_frames = torch.from_numpy(frames).to(gpu)
    res1, st1 = c.detect_batch_device(d_frames, stats=True)
    res2, st2 = c.detect_batch_device(d_frames, stats=True)
    assert st1["patch_n"] == 256 * 38245 == 9790720
    for k in ("cart_gothrough_n", "cart_total_n", "face_patch_n", "stage_done_n", "scan_cart_n", "scan_patch_n"):
        assert st1[k] == st2[k], k                                        # deterministic work
    for a, b in zip(res1, res2):
        _compare_detect(a, b)                                             # deterministic results
    assert 10 < st1["average_cart_n"] < 60                                # the calibrated regime
    # batch entry == host-frame entry == the reference-style single call
    host = c.detect_batch(frames[:8])
    for i in range(8):
        _compare_detect(res1[i], host[i])
        _compare_detect(res1[i], c.detect(frames[i]))
    # oracle / reference on a sample of frames
    o = pyoracle.Oracle(p)
    ref = pyoracle.Reference(p, S_DIMS, 8) if pyoracle.reference_lib_path(*S_DIMS) else None
    for i in (0, 17, 255):
        _compare_detect(res1[i], o.detect(frames[i]))
        if ref is not None:
            _compare_detect(res1[i], ref.detect(frames[i]))
    # counters: carts evaluated over 3 frames equal the oracle's per-window sum
    sub = [0, 100, 255]
    _, st = c.detect_batch(frames[sub], stats=True)
    trs = [o.trace(frames[i], want_shapes=False) for i in sub]
    assert st["cart_total_n"] == sum(int(t["carts_n"].sum()) for t in trs)
    # DetectionStatisic semantics: reject lengths of the non-face windows only
    faces = sum(int(((t["carts_n"] == 2700) & ~(t["score"] < np.float32(-0.5))).sum()) for t in trs)
    assert st["face_patch_n"] == faces and st["nonface_patch_n"] == st["patch_n"] - faces
    assert st["cart_gothrough_n"] == st["cart_total_n"] - faces * 2700
    # NMS off returns a superset, in scan order
    raw = c.detect_batch(frames[:2], nms=False)
    for i in range(2):
        assert len(raw[i]["scores"]) >= len(res1[i]["scores"])
        keep = api.nms_c(raw[i]["bboxes"], raw[i]["scores"])
        assert same(raw[i]["bboxes"][keep], res1[i]["bboxes"])


def test_1080p_eight_levels(built, gpu, model_file):
    """BASELINE.json configs[2] geometry: 1080p, scale 1.5 -> 8 window sizes, 125,350 windows."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 20, 5, 4), 8, seed=5, cart_th=-0.6, norm_every=6)
    frames = synth.make_frames(2, 1920, 1080, seed=21)
    c, o = api.Cascador(p), Oracle(p)
    assert api.count_windows(1920, 1080, 1.5, 40, -1) == (125350, 8)
    _compare_trace(c, o, frames[:1], scale=1.5)
    dets, st = c.detect_batch(frames, scale=1.5, stats=True)
    assert st["patch_n"] == 2 * 125350
    for i in range(2):
        _compare_detect(dets[i], o.detect(frames[i], scale=1.5))


def test_config2_1080p_batch_shipped_dims(built, gpu, tmp_path):
    """BASELINE.json configs[2] with the shipped model dimensions: a 32-frame 1080p batch (66 MB of
    frames), 8 window sizes (scale 1.5), S dims in the calibrated cascade regime.  Properties at full
    size (window accounting, determinism, batch == single-frame call == host-frame entry) and the
    oracle + the compiled reference on sampled frames (c/jda.c:318-480)."""
    import torch
    from jda_amd import api, synth
    from oracle import pyoracle
    frames = synth.make_frames(32, 1920, 1080, seed=0)
    m = synth.make_model(*S_DIMS, seed=1)
    synth.calibrate_thresholds(m, frames[:4], scale=1.5)
    p = str(tmp_path / "cfg2.model"); m.save(p, 8)
    c = api.Cascador(p)
    d_frames = torch.from_numpy(frames).to(gpu)
    kw = dict(scale=1.5)
    res1, st1 = c.detect_batch_device(d_frames, stats=True, **kw)
    res2, st2 = c.detect_batch_device(d_frames, stats=True, **kw)
    assert st1["patch_n"] == 32 * 125350
    for k in ("cart_gothrough_n", "cart_total_n", "face_patch_n", "stage_done_n", "scan_cart_n", "scan_patch_n"):
        assert st1[k] == st2[k], k
    for a, b in zip(res1, res2):
        _compare_detect(a, b)
    assert 10 < st1["average_cart_n"] < 60
    assert sum(len(r["scores"]) for r in res1) > 0
    host = c.detect_batch(frames[:3], **kw)
    for i in range(3):
        _compare_detect(res1[i], host[i])
    _compare_detect(res1[5], c.detect(frames[5], 1.5, 0.1, 40, -1, -0.5))
    o = pyoracle.Oracle(p)
    ref = pyoracle.Reference(p, S_DIMS, 8) if pyoracle.reference_lib_path(*S_DIMS) else None
    for i in (0, 31):
        _compare_detect(res1[i], o.detect(frames[i], **kw))
        if ref is not None:
            _compare_detect(res1[i], ref.detect(frames[i], **kw))
    # carts evaluated over two frames equal the oracle's per-window sum; every window traced
    sub = [7, 20]
    _, st = c.detect_batch(frames[sub], stats=True, **kw)
    trs = [o.trace(frames[i], want_shapes=False, **kw) for i in sub]
    assert st["cart_total_n"] == sum(int(t["carts_n"].sum()) for t in trs)
    _compare_trace(c, o, frames[7:8], **kw)


def test_config2_at_its_stated_size_256_frames_1080p(built, gpu, tmp_path):
    """BASELINE.json configs[2] at the size it states: 256 frames 1920x1080 (531 MB of frames resident in HBM),
    8 window sizes (scale 1.5, c/jda.c:331-333), shipped model dimensions, cascade regime: 32,089,600 candidate
    windows per call.  Size-independent properties over the whole batch (window accounting, determinism, the
    batch equals its halves) and the oracle (+ the compiled reference c/jda.c:318-480) on sampled frames."""
    import json
    import time
    import torch
    from jda_amd import api, synth
    from oracle import pyoracle
    n = 256
    frames = synth.make_frames(n, 1920, 1080, seed=0)
    m = synth.make_model(*S_DIMS, seed=1)
    synth.calibrate_thresholds(m, frames[:4], scale=1.5)
    p = str(tmp_path / "cfg2_full.model"); m.save(p, 8)
    c = api.Cascador(p)
    d_frames = torch.from_numpy(frames).to(gpu)
    kw = dict(scale=1.5)
    res1, st1 = c.detect_batch_device(d_frames, stats=True, **kw)
    assert st1["patch_n"] == n * 125350 == 32089600
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res2, st2 = c.detect_batch_device(d_frames, stats=True, **kw)
    el = time.perf_counter() - t0
    for k in ("cart_gothrough_n", "cart_total_n", "face_patch_n", "stage_done_n", "scan_cart_n", "scan_patch_n"):
        assert st1[k] == st2[k], k
    for a, b in zip(res1, res2):
        _compare_detect(a, b)
    assert 10 < st1["average_cart_n"] < 60
    assert sum(len(r["scores"]) for r in res1) > n
    # additivity: the counters of the batch are the sums over its two halves, the results the concatenation
    ra, sa = c.detect_batch_device(d_frames[:128], stats=True, **kw)
    rb, sb = c.detect_batch_device(d_frames[128:], stats=True, **kw)
    for k in ("cart_total_n", "face_patch_n", "cart_gothrough_n", "handoff_n"):
        assert sa[k] + sb[k] == st1[k], k
    for a, b in zip(ra + rb, res1):
        _compare_detect(a, b)
    o = pyoracle.Oracle(p)
    ref = pyoracle.Reference(p, S_DIMS, 8) if pyoracle.reference_lib_path(*S_DIMS) else None
    for i in (3, 200):
        _compare_detect(res1[i], o.detect(frames[i], **kw))
        if ref is not None:
            _compare_detect(res1[i], ref.detect(frames[i], **kw))
    _compare_detect(res1[255], c.detect(frames[255], 1.5, 0.1, 40, -1, -0.5))
    rec = {"config": "BASELINE.json configs[2]: 256 x 1920x1080, scale 1.5 (8 levels), S dims, cascade regime",
           "windows": st1["patch_n"], "ms_per_call": el * 1e3, "windows_per_s": st1["patch_n"] / el,
           "gpu_ms": st2["gpu_ms"], "average_cart_n": st1["average_cart_n"], "handoff_n": st1["handoff_n"]}
    print("config2 full size:", json.dumps(rec))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "config2_full.json"), "w") as f:
            f.write(json.dumps(rec) + "\n")


X_DIMS = (7, 2000, 68, 6)


@pytest.fixture(scope="module")
def x_model(tmp_path_factory):
    """BASELINE.json configs[4]: T=7, K=2000, 68 landmarks, depth 6 (NODE=31, LEAF=32; c/jda.c:24-27,130-151)
    as a float model file (259,560,576 B; W = 243.7 MB).  cart_th=-2: ~1 % of the windows finish all 14,000
    carts, so k_finish's depth-6 walks and its 34.8 MB-per-stage W gather run for thousands of windows."""
    from jda_amd import synth
    p = str(tmp_path_factory.mktemp("x") / "x.model")
    synth.make_model(*X_DIMS, seed=2, cart_th=-2.0).save(p, 4)
    assert os.path.getsize(p) == 259560576
    return p


def test_config4_deep_model_small_frame(built, gpu, x_model):
    """configs[4] model on 320x240: every window's trace and the detections vs the oracle and the reference."""
    from jda_amd import api, synth
    from oracle import pyoracle
    c, o = api.Cascador(x_model, "float"), pyoracle.Oracle(x_model)
    assert (c.T, c.K, c.L, c.D) == X_DIMS
    frames = synth.make_frames(2, 320, 240, seed=3)
    _compare_trace(c, o, frames)
    dets = c.detect_batch(frames)
    ref = pyoracle.Reference(x_model, X_DIMS, 4) if pyoracle.reference_lib_path(*X_DIMS) else None
    for i in range(2):
        _compare_detect(dets[i], o.detect(frames[i]))
        if ref is not None:
            _compare_detect(dets[i], ref.detect(frames[i]))
    assert sum(len(d["scores"]) for d in dets) > 0


def test_config4_deep_model_1080p(built, gpu, x_model):
    """configs[4] at its stated shape: one 1080p frame, canonical call (303,222 windows / 15 sizes).  Window by
    window vs the oracle (reject position, score bits, leaf-path hash, shape bits) in a regime where a few
    thousand windows finish every stage; detections vs the compiled reference; counters of a 2-frame batch."""
    import torch
    from jda_amd import api, synth
    from oracle import pyoracle
    c, o = api.Cascador(x_model, "float"), pyoracle.Oracle(x_model)
    frames = synth.make_frames(2, 1920, 1080, seed=4)
    assert api.count_windows(1920, 1080, 1.25, 40, -1) == (303222, 15)
    g = c.trace(frames[:1])
    r = o.trace(frames[0])
    for k in ("carts_n", "score", "path_hash", "shapes"):
        assert same(r[k], g[k]), (k, int((bits(r[k]) != bits(g[k])).sum()))
    finished = int((r["carts_n"] == 14000).sum())
    assert finished > 300                                                  # the W gather is exercised
    res, st = c.detect_batch_device(torch.from_numpy(frames).to(gpu), stats=True)
    assert st["patch_n"] == 2 * 303222
    assert st["stage_done_n"][6] >= finished
    one, st1 = c.detect_batch(frames[:1], stats=True)
    assert st1["cart_total_n"] == int(r["carts_n"].sum())
    assert st1["stage_done_n"][6] == finished
    _compare_detect(one[0], res[0])
    _compare_detect(res[0], o.detect(frames[0]))
    if pyoracle.reference_lib_path(*X_DIMS):
        ref = pyoracle.Reference(x_model, X_DIMS, 4)
        _compare_detect(res[0], ref.detect(frames[0]))


def test_concurrent_callers_share_one_cascador(built, gpu, model_file):
    """jdaDetect is re-entrant in the reference (no globals); ours serialises internally."""
    from concurrent.futures import ThreadPoolExecutor
    from jda_amd import api, synth
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-1.0)
    frames = synth.make_frames(8, 200, 150, seed=1)
    c = api.Cascador(p)
    want = [c.detect(f) for f in frames]
    with ThreadPoolExecutor(4) as ex:
        got = list(ex.map(c.detect, list(frames) * 3))
    for i, g in enumerate(got):
        _compare_detect(g, want[i % 8])


# ---------------------------------------------------------------- dialect CPP (src/jda, fp64)

def _compare_trace_cpp(casc, orc, frames, **kw):
    g = casc.trace_cpp(frames, **kw)
    off = 0
    for i in range(len(frames)):
        r = orc.trace_cpp(frames[i], **kw)
        n = len(r["carts_n"])
        for k in ("carts_n", "score", "path_hash", "shapes"):
            assert same(r[k], g[k][off:off + n]), (i, k, int((bits(r[k]) != bits(g[k][off:off + n])).sum()))
        off += n
    assert off == len(g["carts_n"])


@pytest.mark.parametrize("dims", [(2, 8, 5, 3), (3, 20, 5, 4), (2, 6, 4, 6), (3, 70, 9, 5)])
@pytest.mark.parametrize("th", [-3.0e38, -0.8])
def test_dialect_cpp_vs_oracle(built, gpu, model_file, dims, th):
    """fp64 / round() / fixed pixel step / delta-shape summed from zero / score-ordered NMS:
    reference src/jda/cascador.cpp:166-211,310-477 as restated by the oracle (parity unpinned
    against a compiled src/jda -- it needs OpenCV -- see DESIGN.md)."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file(dims, 8, seed=21, cart_th=th, norm_every=5, f32_exact=False)
    frames = synth.make_frames(2, 160, 120, seed=31)
    c, o = api.Cascador(p, "double"), Oracle(p)
    kw = dict(minimum_size=20, step=5, factor=1.2)
    _compare_trace_cpp(c, o, frames, **kw)
    for nms in (True, False):
        dets, st = c.detect_batch_cpp(frames, overlap=0.3, nms=nms, stats=True, **kw)
        assert st["patch_n"] == 2 * synth.count_windows_cpp(160, 120, 20, 5, 1.2)
        for i in range(len(frames)):
            want = o.detect_cpp(frames[i], overlap=0.3, nms=nms, **kw)
            for k in ("rects", "scores", "shapes"):
                assert same(dets[i][k], want[k]), (nms, i, k)


def test_dialect_cpp_shipped_config(built, gpu, model_file):
    """fddb parameters of the shipped config (model/config.json:41-45: min 20, step 5, x1.2)."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 20, 5, 4), 8, seed=22, cart_th=-0.6)
    frames = synth.make_frames(1, 640, 480, seed=32)
    c, o = api.Cascador(p), Oracle(p)
    dets, st = c.detect_batch_cpp(frames, 20, 5, 1.2, 0.3, True, stats=True)
    assert st["patch_n"] == 140215
    want = o.detect_cpp(frames[0], 20, 5, 1.2, 0.3, True)
    for k in want:
        assert same(dets[0][k], want[k]), k


DIFF_KERNELS = {  # which finishing kernel a traced pass runs (options of DESIGN.md section 8; none changes results)
    "k_finish": dict(dense=0, wide_max=0),
    "k_finish_wide": dict(dense=0, wide_max=10_000_000),
    "k_stage": dict(dense=2),
}


@pytest.mark.parametrize("kernel", sorted(DIFF_KERNELS))
@pytest.mark.parametrize("dims,win,scale,n_frames,w,h", [((3, 70, 9, 5), 32, 4.0 / 3.0, 4, 640, 480),
                                                         ((2, 64, 68, 6), 64, 8.0 / 3.0, 2, 400, 300)])
@pytest.mark.parametrize("reject", [0.0, 0.12])
def test_dialects_agree_where_they_must(built, gpu, tmp_path, kernel, dims, win, scale, n_frames, w, h, reject):
    """The fp64 instantiation of the kernels (dialect CPP, parity UNPINNED: src/jda needs OpenCV) against their fp32
    instantiation (dialect C, pinned by the compiled c/jda.c) on a model where the two MUST agree
    (synth.make_dyadic_model: truncation and round() pick the same pixels, every sum is exact in fp32 and fp64) and on
    the same window grid (one window size; C: min = max = win, CPP: minimum_size = win, step = (int)(0.1f * win)).
    Per window: reject position (carts evaluated), leaf-path hash, score and shape after fp64 -> fp32 conversion --
    121,800 windows in the large case, through each of the three finishing kernels.  The CPU half (the oracle's two
    dialects, and this model through the compiled reference) is tests/test_dialect_differential.py.  What stays
    unpinned in dialect CPP: round() where it differs from truncation, fp64 accumulation where fp32 rounds, the patch
    sizes of scale != 0 nodes, cv::resize, the multimap NMS (cart.cpp:392-404, data.cpp:37-51)."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    m = synth.make_dyadic_model(*dims, win=win, seed=11 + win, reject=reject)
    p = str(tmp_path / "dyadic.model")
    m.save(p, 8)
    frames = synth.make_frames(n_frames, w, h, seed=33)
    step = int(np.float32(win) * np.float32(0.1))
    c = api.Cascador(p)
    for k, v in DIFF_KERNELS[kernel].items():
        c.set_option(k, v)
    a = c.trace(frames, scale=np.float32(scale), min_size=win, max_size=win)
    b = c.trace_cpp(frames, minimum_size=win, step=step, factor=100.0)
    n = n_frames * ((w - win) // step + 1) * ((h - win) // step + 1)
    assert len(a["carts_n"]) == len(b["carts_n"]) == n
    if dims == (3, 70, 9, 5):
        assert n >= 100_000
    assert np.array_equal(a["carts_n"], b["carts_n"])
    assert np.array_equal(a["path_hash"], b["path_hash"])
    assert same(a["score"], b["score"].astype(np.float32))
    assert same(a["shapes"], b["shapes"].astype(np.float32))
    assert np.array_equal(b["shapes"].astype(np.float32).astype(np.float64), b["shapes"])
    T, K = dims[0], dims[1]
    if reject == 0.0:
        assert (a["carts_n"] == T * K).all()
    else:
        assert 0.2 < (a["carts_n"] < T * K).mean() and (a["carts_n"] > K).any() and len(np.unique(a["carts_n"])) > 10
    assert len(np.unique(a["path_hash"])) > n // 8
    # ... and the pinned side is the oracle's on the first frame (which the CPU half ties to the compiled reference)
    r = Oracle(p).trace(frames[0], scale=np.float32(scale), min_size=win, max_size=win)
    for key in ("carts_n", "score", "path_hash", "shapes"):
        assert same(r[key], a[key][:len(r[key])]), key


@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 452, 339), (640, 480, 320, 240), (131, 97, 92, 68), (100, 80, 100, 80),
                                         (97, 131, 48, 65), (60, 40, 90, 70), (450, 337, 375, 280), (33, 21, 1, 1)])
def test_cv_resize_kernel_vs_oracle(built, gpu, model_file, sw, sh, dw, dh):
    """Device restatement of cv::resize(INTER_LINEAR) == the oracle's (both UNPINNED against
    OpenCV itself): general bilinear, the exact-2x box path, identity, up-scaling."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((2, 8, 5, 3), 8)
    img = synth.make_frames(1, sw, sh, seed=sw + dh)[0]
    got, want = api.Cascador(p).resize_cv(img, dw, dh), Oracle(p).resize_cv(img, dw, dh)
    assert np.array_equal(got, want)
    if (sw, sh) == (dw, dh):
        assert np.array_equal(got, img)                          # same-size resize is the identity


def test_dialect_cpp_multiscale_method1(built, gpu, model_file):
    """scale != 0 nodes in dialect CPP: half/quarter images by cv::resize on the device, patches
    (int(x/r), int(y/r), int(win/r)) and (x/2, y/2, win/2), coordinates scaled by the PATCH size
    (cascador.cpp:329-353, data.cpp:21-51)."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 20, 5, 4), 8, seed=24, cart_th=-0.9, multi_scale=True, f32_exact=False)
    frames = synth.make_frames(2, 200, 150, seed=34)
    c, o = api.Cascador(p), Oracle(p)
    kw = dict(minimum_size=20, step=5, factor=1.2)
    _compare_trace_cpp(c, o, frames, **kw)
    got = c.detect_batch_cpp(frames, overlap=0.3, nms=True, **kw)
    for i in range(2):
        want = o.detect_cpp(frames[i], overlap=0.3, nms=True, **kw)
        for k in want:
            assert same(got[i][k], want[k]), k
        assert len(want["scores"]) > 0


@pytest.mark.parametrize("size,origin,step,factor", [((200, 150), 48, 5, 1.2), ((131, 97), 24, 3, 1.3), ((320, 240), 48, 8, 1.5)])
def test_dialect_cpp_method0_true_pyramid(built, gpu, model_file, size, origin, step, factor):
    """detectMultiScale + detectSingleScale (cascador.cpp:216-308): pyramid levels built on the
    device by repeated cv::resize, fixed window, rects scaled back with truncation."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 20, 5, 4), 8, seed=25, cart_th=-0.7, norm_every=6, f32_exact=False)
    frames = synth.make_frames(3, size[0], size[1], seed=35)
    c, o = api.Cascador(p), Oracle(p)
    for nms in (True, False):
        got, st = c.detect_batch_cpp_pyramid(frames, origin, step, factor, 0.3, nms, stats=True)
        tot = 0
        for i in range(len(frames)):
            want = o.detect_cpp_pyramid(frames[i], origin, step, factor, 0.3, nms)
            for k in ("rects", "scores", "shapes"):
                assert same(got[i][k], want[k]), (nms, i, k)
            tot += want["windows"]
        assert st["patch_n"] == tot
    assert sum(len(g["scores"]) for g in got) > 0


def test_method0_multiscale_needs_the_patch_sizes(built, gpu, model_file):
    from jda_amd import api, synth
    p, _ = model_file((2, 8, 5, 3), 8, seed=24, multi_scale=True)
    c = api.Cascador(p)
    with pytest.raises(api.JdaError, match="half_size and quarter_size"):
        c.detect_batch_cpp_pyramid(synth.make_frames(1, 64, 64, seed=1))


@pytest.mark.parametrize("dims,sizes,frame", [((3, 20, 5, 4), (48, 36, 24), (160, 120)), ((2, 8, 5, 3), (40, 27, 20), (131, 97)),
                                              ((3, 70, 9, 5), (48, 36, 24), (200, 150))])
def test_method0_multiscale_per_window_patches(built, gpu, model_file, dims, sizes, frame):
    """Method 0 on a model with scale != 0 split nodes: detectSingleScale resizes every window's ROI to the config's
    half_size / quarter_size (cascador.cpp:243-245; 48 / 36 / 24 in the shipped config, the exact-2x quarter goes through
    cv::resize's box average) and a split node reads the window's own patch with coordinates scaled by that patch's side
    (data.cpp:21-51).  The patches are built on the device; detections == the oracle's (both UNPINNED: src/jda needs OpenCV)."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file(dims, 8, seed=24, cart_th=-0.9, multi_scale=True, f32_exact=False)
    o_size, h_size, q_size = sizes
    frames = synth.make_frames(2, frame[0], frame[1], seed=35)
    c, o = api.Cascador(p), Oracle(p)
    total = 0
    for nms in (True, False):
        got, st = c.detect_batch_cpp_pyramid(frames, o_size, 5, 1.2, 0.3, nms, stats=True, half_size=h_size, quarter_size=q_size)
        for i in range(len(frames)):
            want = o.detect_cpp_pyramid(frames[i], o_size, 5, 1.2, 0.3, nms, half_size=h_size, quarter_size=q_size)
            assert st["patch_n"] == 2 * want["windows"]
            for k in ("rects", "scores", "shapes"):
                assert same(got[i][k], want[k]), (nms, i, k, len(got[i]["scores"]), len(want["scores"]))
            total += len(want["scores"])
    assert total > 0
    # a single-scale model through the same entry: the sizes are unused, results those of jdaDetectBatchCppPyramid
    p1, _ = model_file(dims, 8, seed=25, cart_th=-0.9)
    c1 = api.Cascador(p1)
    a = c1.detect_batch_cpp_pyramid(frames, o_size, 5, 1.2)
    b = c1.detect_batch_cpp_pyramid(frames, o_size, 5, 1.2, half_size=h_size, quarter_size=q_size)
    for x, y in zip(a, b):
        for k in ("rects", "scores", "shapes"):
            assert same(x[k], y[k]), k


# ---------------------------------------------------------------- randomised sweep

def test_random_configurations_vs_oracle(built, gpu, tmp_path):
    """40 random (dims, frame size, call arguments, threshold) combinations, both dialects:
    detections and per-window traces bit-exact against the oracle."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    rng = np.random.default_rng(2024)
    for case in range(40):
        T = int(rng.integers(1, 5)); K = int(rng.choice([1, 3, 17, 64, 65, 130, 200])); L = int(rng.integers(1, 12))
        D = int(rng.integers(2, 7))
        th = float(rng.choice([-3.0e38, -2.0, -0.9, -0.2]))
        m = synth.make_model(T, K, L, D, seed=100 + case, cart_th=th, norm_every=int(rng.integers(2, 9)),
                             multi_scale=bool(rng.random() < 0.2), f32_exact=bool(rng.random() < 0.5))
        p = str(tmp_path / ("r%d.model" % case)); m.save(p, 8 if rng.random() < 0.5 else 4)
        w, h = int(rng.integers(24, 330)), int(rng.integers(24, 260))
        frames = synth.make_frames(int(rng.integers(1, 4)), w, h, seed=case)
        kw = dict(scale=float(rng.choice([1.1, 1.25, 1.5, 2.0])), min_size=int(rng.integers(0, 60)),
                  max_size=int(rng.choice([-1, 0, 50, 120])))
        c, o = api.Cascador(p), Oracle(p)
        _compare_trace(c, o, frames, **kw)
        fin = float(rng.choice([-10.0, -0.5, 0.3]))
        dets = c.detect_batch(frames, th=fin, **kw)
        for i in range(len(frames)):
            _compare_detect(dets[i], o.detect(frames[i], th=fin, **kw))
        if True:
            ckw = dict(minimum_size=int(rng.integers(12, 40)), step=int(rng.integers(2, 9)),
                       factor=float(rng.choice([1.15, 1.2, 1.5])))
            if min(w, h) >= ckw["minimum_size"]:
                _compare_trace_cpp(c, o, frames, **ckw)
                got = c.detect_batch_cpp(frames, overlap=0.3, nms=True, **ckw)
                for i in range(len(frames)):
                    want = o.detect_cpp(frames[i], overlap=0.3, nms=True, **ckw)
                    for k in want:
                        assert same(got[i][k], want[k]), (case, k)
        c.close()


# ---------------------------------------------------------------- sizes, strides, passes

def test_4k_frame_and_odd_strides(built, gpu, model_file):
    """3840x2160 (window offsets no longer fit the packed stage-0 node for the largest levels ->
    those enter k_finish at cart 0), a padded frame stride and a device pointer that is not
    4-byte aligned (byte-wise tile staging)."""
    import ctypes as C
    import torch
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 24, 5, 4), 8, seed=41, cart_th=0.1, norm_every=5)
    c, o = api.Cascador(p), Oracle(p)
    big = synth.make_frames(1, 3840, 2160, seed=42)
    _compare_trace(c, o, big, scale=1.6, min_size=48)
    d = c.detect_batch(big, scale=1.6, min_size=48, th=0.6)[0]
    _compare_detect(d, o.detect(big[0], scale=1.6, min_size=48, th=0.6))
    assert len(d["scores"]) > 50
    # padded stride + misaligned base: frames live inside a larger byte buffer at offset 1
    fr = synth.make_frames(3, 200, 150, seed=43)
    stride = 200 * 150 + 37
    flat = torch.zeros(1 + 3 * stride, dtype=torch.uint8, device=gpu)
    for i in range(3):
        flat[1 + i * stride: 1 + i * stride + 200 * 150] = torch.from_numpy(fr[i].reshape(-1)).to(gpu)
    res = (api.jdaResult * 3)()
    opt = api.jdaDetectOptions(); api.lib.jdaDetectOptionsInit(C.byref(opt))
    rc = api.lib.jdaDetectBatchDevice(c.h, C.c_void_p(flat.data_ptr() + 1), stride, 3, 200, 150, 1.25, 0.1, 40, -1, -0.5,
                                      C.byref(opt), res)
    assert rc == 0, api.last_error()
    for i in range(3):
        _compare_detect(api._take(res[i]), o.detect(fr[i]))


def test_very_wide_frame_stage0_offsets_beyond_21_bits(built, gpu, model_file):
    """24000x110: the LDS-tiled levels' window offsets inside the FRAME ((win-1)*W + win-1) exceed 21 bits (the packing
    of the global-pixel scan's table); k_finish's stage-0 table holds (x, y) pairs instead and k_scan its tile-relative
    offsets, so both still take their table paths."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 24, 5, 4), 8, seed=45, cart_th=-0.2, norm_every=5)
    c, o = api.Cascador(p), Oracle(p)
    wide = synth.make_frames(1, 24000, 110, seed=46)
    levels = c.plan_tiles(24000, 110, 1.1, 80, -1)
    assert any(lv["mode"] in (1, 3) and (lv["win"] - 1) * 24000 + lv["win"] - 1 >= 2 ** 21 for lv in levels)
    _compare_trace(c, o, wide, scale=1.1, min_size=80)
    d = c.detect_batch(wide, scale=1.1, min_size=80, th=0.0)[0]
    _compare_detect(d, o.detect(wide[0], scale=1.1, min_size=80, th=0.0))


def test_batch_larger_than_workspace_is_processed_in_passes(built, gpu, model_file, monkeypatch):
    from jda_amd import api, synth
    p, _ = model_file((3, 20, 5, 4), 8, seed=44, cart_th=-0.8)
    frames = synth.make_frames(9, 240, 180, seed=45)
    want = api.Cascador(p).detect_batch(frames)
    monkeypatch.setenv("JDA_WORKSPACE_MB", "1")            # ~2 frames per pass
    c = api.Cascador(p)
    got, st = c.detect_batch(frames, stats=True)
    assert st["patch_n"] == 9 * api.count_windows(240, 180)[0]
    for a, b in zip(got, want):
        _compare_detect(a, b)
    tr1 = c.trace(frames)
    monkeypatch.delenv("JDA_WORKSPACE_MB")
    tr2 = api.Cascador(p).trace(frames)
    for k in tr1:
        assert same(tr1[k], tr2[k]), k


@pytest.mark.parametrize("dims", [(3, 20, 5, 4), (2, 70, 9, 5), (4, 12, 27, 3)])
def test_dialect_cpp_similarity_transform(built, gpu, model_file, dims):
    """face.similarity_transform = true (data.cpp:64-126): per-stage sR = Calc(shape, mean_shape)
    applied to node offsets and to the regressed delta; methods 1 and 0.  Unpinned like the rest of
    dialect CPP (plus cv::norm / Mat_ /= details), bit-exact against the oracle's restatement."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file(dims, 8, seed=51, cart_th=-0.8, norm_every=5, f32_exact=False, w_sigma=6e-3)
    frames = synth.make_frames(2, 160, 120, seed=52)
    c, o = api.Cascador(p), Oracle(p)
    kw = dict(minimum_size=20, step=5, factor=1.2)
    base = c.trace_cpp(frames, **kw)
    try:
        o.set_similarity_transform(True)
        c.set_similarity_transform(True)
        _compare_trace_cpp(c, o, frames, **kw)
        on = c.trace_cpp(frames, **kw)
        assert not same(on["shapes"], base["shapes"])               # the mode really changes results
        got = c.detect_batch_cpp(frames, overlap=0.3, nms=True, **kw)
        pyr = c.detect_batch_cpp_pyramid(frames, 24, 4, 1.25, 0.3, True)
        for i in range(2):
            want = o.detect_cpp(frames[i], overlap=0.3, nms=True, **kw)
            wp = o.detect_cpp_pyramid(frames[i], 24, 4, 1.25, 0.3, True)
            for k in ("rects", "scores", "shapes"):
                assert same(got[i][k], want[k]), k
                assert same(pyr[i][k], wp[k]), k
        c.set_similarity_transform(False)
        o.set_similarity_transform(False)
        _compare_trace_cpp(c, o, frames, **kw)                      # and switching back restores the default
        back = c.trace_cpp(frames, **kw)
        for k in base:
            assert same(back[k], base[k]), k
    finally:
        o.set_similarity_transform(False)


def test_plan_cache_is_bounded(built, gpu, model_file, monkeypatch):
    """A stream of differently sized images (the FDDB case) evicts old scan plans instead of
    accumulating device tables; results are unaffected by eviction and re-creation."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    monkeypatch.setenv("JDA_PLAN_CACHE", "3")
    p, _ = model_file((3, 20, 5, 4), 8, seed=61, cart_th=-0.9)
    c, o = api.Cascador(p), Oracle(p)
    sizes = [(120 + 7 * i, 100 + 5 * i) for i in range(8)]
    imgs = [synth.make_frames(1, w, h, seed=i)[0] for i, (w, h) in enumerate(sizes)]
    for rep in range(2):
        for im in imgs + imgs[::-1]:
            _compare_detect(c.detect(im), o.detect(im))


# ---------------------------------------------------------------- dense mode (k_stage)

@pytest.mark.parametrize("dims,cart_th", [((3, 70, 9, 5), None), ((3, 70, 9, 5), -0.4), ((2, 130, 27, 4), -0.2),
                                          ((4, 12, 40, 3), None)])
def test_dense_mode_is_a_third_implementation_of_the_walk(built, gpu, model_file, monkeypatch, dims, cart_th):
    """Tile-per-workgroup whole-stage kernel vs the scan + wave-per-window pipeline vs the oracle:
    same carts_n / score / hash / shape bits, same detections, same counters.  `auto` picks it when most
    windows survive the scan; `2` forces it even for a rejecting cascade (tiles die off early)."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    kw = {} if cart_th is None else {"cart_th": cart_th, "norm_every": 7}
    p, _ = model_file(dims, 8, seed=61, **kw)
    frames = synth.make_frames(3, 333, 250, seed=62)
    o = Oracle(p)
    monkeypatch.setenv("JDA_DENSE", "0")
    c0 = api.Cascador(p)
    tr0 = c0.trace(frames)
    d0, s0 = c0.detect_batch(frames, stats=True)
    assert s0["dense_passes"] == 0
    for mode in ("2", "1"):
        monkeypatch.setenv("JDA_DENSE", mode)
        c = api.Cascador(p)
        _compare_trace(c, o, frames[:1])
        tr = c.trace(frames)
        for k in tr0:
            assert same(tr0[k], tr[k]), (mode, k)
        d, s = c.detect_batch(frames, stats=True)
        for a, b in zip(d, d0):
            _compare_detect(a, b)
        for k in ("patch_n", "face_patch_n", "cart_gothrough_n", "cart_total_n"):
            assert s[k] == s0[k], (mode, k)
        assert list(s["stage_done_n"]) == list(s0["stage_done_n"])
        if mode == "2" or cart_th is None:
            assert s["dense_passes"] >= 1, "dense mode did not run"


def test_dense_mode_dialect_cpp(built, gpu, model_file, monkeypatch):
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((3, 20, 5, 4), 8, seed=63)
    frames = synth.make_frames(2, 200, 150, seed=64)
    monkeypatch.setenv("JDA_DENSE", "2")
    c, o = api.Cascador(p), Oracle(p)
    g = c.trace_cpp(frames, 20, 5, 1.2)
    off = 0
    for i in range(len(frames)):
        r = o.trace_cpp(frames[i], 20, 5, 1.2)
        n = len(r["carts_n"])
        for k in ("carts_n", "score", "path_hash", "shapes"):
            assert same(r[k], g[k][off:off + n]), (i, k)
        off += n


# ---------------------------------------------------------------- two lanes

@pytest.mark.parametrize("passes", [False, True])
def test_two_lanes_give_the_same_results_as_one(built, gpu, model_file, monkeypatch, passes):
    """Big batches are split into sub-batches that run on two streams with their own workspace
    (run_device); forced here on a small batch, also with more sub-batches than lanes."""
    from jda_amd import api, synth
    p, _ = model_file((3, 70, 9, 5), 8, seed=71, cart_th=-0.9, norm_every=9)
    frames = synth.make_frames(7, 240, 180, seed=72)
    monkeypatch.setenv("JDA_LANES", "1")
    c1 = api.Cascador(p)
    want, s1 = c1.detect_batch(frames, stats=True)
    tr1 = c1.trace(frames)
    monkeypatch.setenv("JDA_LANES", "2")
    monkeypatch.setenv("JDA_LANES_MIN_WINDOWS", "1")
    if passes:
        monkeypatch.setenv("JDA_WORKSPACE_MB", "1")        # ~1 frame per sub-batch: several rounds
    c2 = api.Cascador(p)
    got, s2 = c2.detect_batch(frames, stats=True)
    for a, b in zip(got, want):
        _compare_detect(a, b)
    for k in ("patch_n", "face_patch_n", "cart_gothrough_n", "cart_total_n", "handoff_n"):
        assert s1[k] == s2[k], k
    assert list(s1["stage_done_n"]) == list(s2["stage_done_n"])
    tr2 = c2.trace(frames)
    for k in tr1:
        assert same(tr1[k], tr2[k]), k
    # a caller-provided stream carries lane 0; the other lane is ordered behind it
    import torch
    st = torch.cuda.Stream()
    d = torch.from_numpy(frames).cuda()
    with torch.cuda.stream(st):
        d2 = d.clone()
        res = c2.detect_batch_device(d2, 1.25, 40, -1, -0.5, hip_stream=st.cuda_stream)
    ref = c1.detect_batch(frames, 1.25, 40, -1, -0.5)
    for a, b in zip(res, ref):
        _compare_detect(a, b)
    # ... also when the plan (stage-0 tables, built on an internal stream) is new in that very call
    c3 = api.Cascador(p)
    with torch.cuda.stream(st):
        res3 = c3.detect_batch_device(d2, 1.25, 40, -1, -0.5, hip_stream=st.cuda_stream)
    for a, b in zip(res3, ref):
        _compare_detect(a, b)


@pytest.mark.parametrize("size", [(100, 75), (64, 50), (131, 58), (90, 200)])
@pytest.mark.parametrize("cart_th", [None, -1.6, -1.0])
def test_levels_with_a_handful_of_windows(built, gpu, model_file, size, cart_th):
    """Tiles with 1..16 windows go through k_scan's late-phase mapping ((window, cart) pairs over all
    lanes, 16-cart register replay) from cart 16 on; with slowly rejecting thresholds they die there."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    kw = {} if cart_th is None else {"cart_th": cart_th, "norm_every": 50}
    p, _ = model_file((2, 160, 9, 4), 8, seed=81, **kw)
    frames = synth.make_frames(3, size[0], size[1], seed=82)
    c, o = api.Cascador(p), Oracle(p)
    _compare_trace(c, o, frames, scale=1.25, min_size=40, max_size=-1)
    for i, d in enumerate(c.detect_batch(frames, 1.25, 40, -1, -1.0)):
        _compare_detect(d, o.detect(frames[i], 1.25, 40, -1, -1.0))


def test_two_lanes_dialect_cpp_and_dense_mode(built, gpu, model_file, monkeypatch):
    """The lane split is dialect-agnostic and composes with the dense path and the side stream."""
    from jda_amd import api, synth
    p, _ = model_file((3, 70, 9, 5), 8, seed=91, cart_th=-0.7, norm_every=11)
    frames = synth.make_frames(6, 220, 170, seed=92)
    monkeypatch.setenv("JDA_LANES", "1")
    monkeypatch.setenv("JDA_SIDE_STREAM", "0")
    c1 = api.Cascador(p)
    want = c1.detect_batch_cpp(frames, 20, 5, 1.2)
    tr1 = c1.trace_cpp(frames, 20, 5, 1.2)
    for env in ({"JDA_LANES": "2", "JDA_LANES_MIN_WINDOWS": "1"},
                {"JDA_LANES": "2", "JDA_LANES_MIN_WINDOWS": "1", "JDA_DENSE": "2"},
                {"JDA_LANES": "1", "JDA_SIDE_STREAM": "1", "JDA_SIDE_SMALL": "1", "JDA_MERGE_BLOCKS": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c2 = api.Cascador(p)
        got = c2.detect_batch_cpp(frames, 20, 5, 1.2)
        for a, b in zip(got, want):
            for k in ("rects", "scores", "shapes"):
                assert same(a[k], b[k]), (env, k)
        tr2 = c2.trace_cpp(frames, 20, 5, 1.2)
        for k in tr1:
            assert same(tr1[k], tr2[k]), (env, k)
        for k in env:
            monkeypatch.delenv(k)
        monkeypatch.setenv("JDA_SIDE_STREAM", "0")


def test_bench_control_flow_with_two_ranks_on_one_gpu(built, gpu):
    """bench.py's N>1 path (barriers, max-over-ranks timing, gather of the detections on rank 0) with two
    ranks that share this GPU over a gloo group -- RCCL itself needs one GPU per rank, the driver runs that."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, JDA_BENCH_BACKEND="gloo", JDA_BENCH_ONE_GPU="1", JDA_DENSE="1")   # the roofline leg needs the scan
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    # no launcher: `python bench.py --gpus 2` starts its two ranks itself (bench.py:maybe_self_spawn), the way the
    # driver invokes it
    cmd = [sys.executable, os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "32", "--no-cpu", "--no-allpass"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["roofline"]["achieved"] > 0 and d["regimes"]["cascade"]["detections_after_nms"] > 0
    cfg = d["config"]
    assert "gather" in cfg and cfg["fddb_images_per_s"] > 0
    # the first-contact self-test of the gather ran before the timed region (an empty rank, an overflowing rank)
    assert cfg["dist_selftest"].startswith("ok"), cfg["dist_selftest"]
    assert 0 < cfg["rank_ms_per_step_min"] <= cfg["rank_ms_per_step_max"] and cfg["rank_ms_per_step_max"] == pytest.approx(d["ms_per_step"])


def test_bench_plain_invocation_one_gpu(built, gpu):
    """`python bench.py --gpus 1 --steps K --warmup W` exactly as the driver runs it (no launcher): one JSON line
    with the contract's keys, the roofline and the FDDB-shaped images/s."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--batch", "32", "--no-cpu", "--no-allpass"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0
    assert 0 < d["roofline"]["frac"] <= 1
    # the whole BASELINE metric sits in the two objects the driver's record keeps: FDDB images/sec, configs[2] measured
    # in this run, the regime in which memory traffic is the bound -- as flat scalars
    cfg, roof = d["config"], d["roofline"]
    assert cfg["fddb_images_per_s"] > 0 and cfg["fddb_images"] == 2845 and d["fddb"]["images"] == 2845
    assert cfg["config2_windows_per_s"] > 1e9 and cfg["config2_windows_per_call"] == 32089600 and cfg["config2_frames"] == 256
    assert cfg["config2_submit_wait_windows_per_s"] > 1e9
    for k, v in list(cfg.items()) + list(roof.items()):
        assert v is None or isinstance(v, (int, float, str, bool)), (k, type(v))


def test_pipelined_gather_device_path_over_rccl_group_of_one(built, gpu):
    """The device side of PipelinedGather (RCCL all_gather issued asynchronously, collected a step later through
    pinned memory) in a process group of one rank -- all a 1-GPU box can host."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="%d", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from jda_amd import dist as jd
pg = jd.PipelinedGather(16, 7, device=dev, force=True)
outs = []
for step in range(4):
    n = [3, 0, 40, 16][step]                      # empty, overflowing (fallback) and exactly full blocks
    m = np.full((n, 7), step + 1, np.float32); m[:, 0] = np.arange(n)
    outs.append(pg.start(m))
outs.append(pg.drain())
assert outs[0] is None
for step, g in enumerate(outs[1:]):
    n = [3, 0, 40, 16][step]
    assert g.shape == (n, 7) and (g[:, 1:] == step + 1).all() and list(g[:, 0]) == list(range(n)), (step, g.shape)
dist.barrier(); dist.destroy_process_group()
print("OK")
''' % (root, port)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


# ---------------------------------------------------------------- small jobs: workgroup = window

@pytest.mark.parametrize("dims,kw", [((3, 70, 9, 5), dict(cart_th=-0.9, norm_every=9)), ((2, 130, 27, 4), dict(cart_th=-0.6)),
                                     ((2, 600, 5, 3), dict(cart_th=-1.5, norm_every=50)), ((3, 20, 150, 4), dict(cart_th=-0.8)),
                                     ((2, 64, 68, 6), dict(cart_th=-0.7))])
def test_wide_finish_equals_wave_finish_and_the_oracle(built, gpu, model_file, dims, kw):
    """Small jobs finish with one WORKGROUP per window (k_finish_wide: thread = cart, weight rows through LDS), large
    ones with one wave per window (k_finish).  Same per-window trace (reject cart, score bits, leaf-path hash, shape
    bits) from both, equal to the oracle's (c/jda.c:357-426; Validate, cascador.cpp:166-211), in both dialects."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file(dims, 8, seed=91, **kw)
    frames = synth.make_frames(2, 260, 190, seed=92)
    o = Oracle(p)
    wide, wave = api.Cascador(p), api.Cascador(p)
    wave.set_option("wide_max", 0)
    assert wide.get_option("wide_max") > 0
    _compare_trace(wide, o, frames[:1])
    tw, tv = wide.trace(frames), wave.trace(frames)
    for k in tw:
        assert same(tw[k], tv[k]), k
    # the wide kernel's two forms: replay and regression side by side (wide_conc = 1, the default) or one after the other
    seq = api.Cascador(p)
    seq.set_option("wide_conc", 0)
    assert wide.get_option("wide_conc") == 1
    ts, tsc = seq.trace(frames), seq.trace_cpp(frames, 20, 5, 1.2)
    for k in tw:
        assert same(tw[k], ts[k]), ("wide_conc", k)
    for a, b in zip(wide.detect_batch(frames), seq.detect_batch(frames)):
        _compare_detect(a, b)
    (dw, sw), (dv, sv) = wide.detect_batch(frames, stats=True), wave.detect_batch(frames, stats=True)
    for a, b in zip(dw, dv):
        _compare_detect(a, b)
    for k in ("patch_n", "face_patch_n", "cart_gothrough_n", "cart_total_n", "handoff_n"):
        assert sw[k] == sv[k], k
    assert list(sw["stage_done_n"]) == list(sv["stage_done_n"])
    assert sum(len(d["scores"]) for d in dw) > 0
    # dialect CPP (fp64: the rows go through registers instead of LDS-DMA, the delta is summed from zero)
    cw, cv = wide.trace_cpp(frames, 20, 5, 1.2), wave.trace_cpp(frames, 20, 5, 1.2)
    for k in cw:
        assert same(cw[k], cv[k]), ("cpp", k)
        assert same(cw[k], tsc[k]), ("cpp wide_conc", k)
    r = o.trace_cpp(frames[0], 20, 5, 1.2)
    for k in ("carts_n", "score", "path_hash", "shapes"):
        assert same(r[k], cw[k][:len(r[k])]), ("cpp oracle", k)
    for a, b in zip(wide.detect_batch_cpp(frames), wave.detect_batch_cpp(frames)):
        for k in ("rects", "scores", "shapes"):
            assert same(a[k], b[k]), ("cpp", k)


@pytest.mark.parametrize("dims,kw", [((2, 64, 68, 6), dict(cart_th=-0.7)), ((3, 40, 9, 7), dict(cart_th=-0.9, norm_every=7)),
                                     ((3, 70, 27, 4), dict(cart_th=-0.8))])
def test_finish_table_layouts_do_not_change_results(built, gpu, model_file, monkeypatch, dims, kw):
    """k_finish reads the regression weights from a copy whose rows start on 128-byte lines (w_pad), with non-temporal
    loads where a stage's rows exceed w_stream_mb, and the last levels of trees with five or more node levels from
    records grouped per path (lm_deep; jda_amd/csrc/kernels.h: lm_deep_index).  None of it may change a bit: per-window
    trace and detections under every combination equal the plain layouts' and the oracle's (c/jda.c:357-426), in both
    dialects (the options are read when a cascador is created)."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file(dims, 8, seed=93, **kw)
    frames = synth.make_frames(2, 260, 190, seed=94)
    monkeypatch.setenv("JDA_W_PAD", "0"); monkeypatch.setenv("JDA_LM_DEEP", "0"); monkeypatch.setenv("JDA_W_STREAM_MB", "0")
    plain = api.Cascador(p)
    plain.set_option("wide_max", 0)            # the wave-per-window kernel is the one that reads these tables
    _compare_trace(plain, Oracle(p), frames[:1])
    want_t, want_d, want_c = plain.trace(frames), plain.detect_batch(frames), plain.trace_cpp(frames, 20, 5, 1.2)
    for pad, deep in ((1, 1), (1, 0), (0, 1)):
        monkeypatch.setenv("JDA_W_PAD", str(pad)); monkeypatch.setenv("JDA_LM_DEEP", str(deep))
        c = api.Cascador(p)
        c.set_option("wide_max", 0)
        t, d, tc = c.trace(frames), c.detect_batch(frames), c.trace_cpp(frames, 20, 5, 1.2)
        for k in want_t:
            assert same(want_t[k], t[k]), (pad, deep, k)
        for a, b in zip(want_d, d):
            _compare_detect(a, b)
        for k in want_c:
            assert same(want_c[k], tc[k]), (pad, deep, "cpp", k)
        c.close()
    plain.close()


def test_streamed_weight_rows_give_the_same_detections(built, gpu, model_file, monkeypatch):
    """The non-temporal form of the row loads (STREAM instantiation of k_finish) on a model small enough to check against
    the oracle: forced by a threshold of one megabyte (300 carts x 32 leaves x 640-byte rows = 6.1 MB per stage)."""
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle
    p, _ = model_file((2, 300, 68, 6), 8, seed=95, cart_th=-0.6)
    frames = synth.make_frames(2, 200, 150, seed=96)
    monkeypatch.setenv("JDA_W_STREAM_MB", "1")            # 300 x 32 leaves x 640 B = 6.1 MB per stage > 1 MB
    c = api.Cascador(p)
    c.set_option("wide_max", 0)
    o = Oracle(p)
    got = c.detect_batch(frames)
    for i in range(len(frames)):
        _compare_detect(o.detect(frames[i]), got[i])
    assert sum(len(g["scores"]) for g in got) > 0
    c.close()


# ---------------------------------------------------------------- submit / wait

def test_submit_wait_gives_the_synchronous_results(built, gpu, model_file):
    """jdaDetectBatchSubmit / jdaDetectBatchWait: two batches in flight from one thread, any collection order,
    same detections and counters as jdaDetectBatchDevice."""
    import torch
    from jda_amd import api, synth
    p, _ = model_file((3, 70, 9, 5), 8, seed=101, cart_th=-0.9, norm_every=9)
    fa = torch.from_numpy(synth.make_frames(5, 240, 180, seed=102)).cuda()
    fb = torch.from_numpy(synth.make_frames(3, 240, 180, seed=103)).cuda()
    fc = torch.from_numpy(synth.make_frames(2, 131, 97, seed=104)).cuda()       # another plan, smaller workspace need
    c = api.Cascador(p)
    want = {}
    for name, f in (("a", fa), ("b", fb), ("c", fc)):
        want[name] = c.detect_batch_device(f, stats=True)
    def check(got, name):
        res, st = got
        for x, y in zip(res, want[name][0]):
            _compare_detect(x, y)
        for k in ("patch_n", "face_patch_n", "cart_gothrough_n", "cart_total_n", "handoff_n"):
            assert st[k] == want[name][1][k], (name, k)
    ta = c.submit_batch_device(fa)
    tb = c.submit_batch_device(fb)
    ta2 = c.submit_batch_device(fa)
    assert {ta, tb, ta2} == {0, 1, 2}
    with pytest.raises(api.JdaError):                       # every ticket in use
        c.submit_batch_device(fc)
    check(c.detect_batch_device(fa, stats=True), "a")       # a synchronous call next to the pending tickets (its own lane)
    check(c.wait_batch(tb, stats=True), "b")                # collected out of order
    tc = c.submit_batch_device(fc)
    check(c.wait_batch(ta, stats=True), "a")
    check(c.wait_batch(tc, stats=True), "c")
    check(c.wait_batch(ta2, stats=True), "a")
    assert api.lib.jdaDetectBatchWait(c.h, 0, None, (api.jdaResult * 1)()) != 0      # nothing pending in that slot any more
    # a stream of batches, one ahead
    t = c.submit_batch_device(fa)
    for i in range(4):
        nxt = c.submit_batch_device(fb if i % 2 == 0 else fa) if i < 3 else None
        check(c.wait_batch(t, stats=True), "a" if i % 2 == 0 else "b")
        t = nxt
    # ... and two ahead (three tickets in flight)
    names = ["a", "b", "c", "a", "c", "b"]
    src = {"a": fa, "b": fb, "c": fc}
    q = [c.submit_batch_device(src[names[0]]), c.submit_batch_device(src[names[1]])]
    for i in range(len(names)):
        if i + 2 < len(names):
            q.append(c.submit_batch_device(src[names[i + 2]]))
        check(c.wait_batch(q.pop(0), stats=True), names[i])
    check(c.detect_batch_device(fa, stats=True), "a")      # and the synchronous path works again afterwards


def test_submit_wait_all_pass_model_takes_the_dense_path(built, gpu, model_file):
    import torch
    from jda_amd import api, synth
    p, _ = model_file((2, 40, 5, 4), 8, seed=105)
    f = torch.from_numpy(synth.make_frames(40, 200, 150, seed=106)).cuda()
    c = api.Cascador(p)
    want, sw = c.detect_batch_device(f, stats=True)
    t = c.submit_batch_device(f)
    got, sg = c.wait_batch(t, stats=True)
    for a, b in zip(got, want):
        _compare_detect(a, b)
    assert sg["cart_total_n"] == sw["cart_total_n"] and sg["dense_passes"] >= 1


def test_submit_host_frames_pageable_and_pinned(built, gpu, model_file):
    """jdaDetectBatchSubmitHost: frames in host memory staged per ticket (pageable: the copy blocks the submit;
    pinned: asynchronous), two batches in flight, same detections as the synchronous host-frame entry."""
    import torch
    from jda_amd import api, synth
    p, _ = model_file((3, 70, 9, 5), 8, seed=111, cart_th=-0.9, norm_every=9)
    frames = synth.make_frames(8, 240, 180, seed=112)
    c = api.Cascador(p)
    want = c.detect_batch(frames)
    pinned = torch.from_numpy(frames[4:]).pin_memory()
    ta = c.submit_batch_host(frames[:4])
    tb = c.submit_batch_host(pinned.numpy())
    got = c.wait_batch(ta) + c.wait_batch(tb)
    assert len(got) == 8
    for x, y in zip(got, want):
        _compare_detect(x, y)
    # a stream of host batches, one ahead, alternating with a device-resident one
    d = torch.from_numpy(frames[:4]).cuda()
    t = c.submit_batch_host(frames[4:])
    t2 = c.submit_batch_device(d)
    for x, y in zip(c.wait_batch(t), want[4:]):
        _compare_detect(x, y)
    for x, y in zip(c.wait_batch(t2), want[:4]):
        _compare_detect(x, y)


@pytest.mark.parametrize("h2d_stream,kernel_d2h", [(1, 1), (0, 0), (2, 1), (3, 0)])
def test_host_frame_stream_upload_and_copy_out_modes(built, gpu, model_file, h2d_stream, kernel_d2h):
    """The host-frame stream under every upload mode (own stream + host wait / lane stream / own stream + device wait /
    + event wait) and both result paths (copy-out kernel, copy engine): three tickets in flight over batches large
    enough to take the upload stream (>= h2d_min_bytes), pinned and pageable, always the synchronous call's results."""
    import torch
    from jda_amd import api, synth
    p, _ = model_file((3, 70, 9, 5), 8, seed=121, cart_th=-0.9, norm_every=9)
    frames = [synth.make_frames(24, 320, 240, seed=122 + j) for j in range(3)]          # 1.8 MB per batch
    c = api.Cascador(p)
    c.set_option("h2d_stream", h2d_stream); c.set_option("kernel_d2h", kernel_d2h)
    c.set_option("h2d_min_bytes", 1 << 20)
    assert c.get_option("h2d_stream") == h2d_stream
    want = [c.detect_batch(f, stats=True) for f in frames]
    pins = [torch.from_numpy(f).pin_memory() for f in frames]
    for src in (frames, [t.numpy() for t in pins]):
        order = [0, 1, 2, 1, 0, 2, 2]
        q = [c.submit_batch_host(src[order[0]]), c.submit_batch_host(src[order[1]], stats=True)]
        for i, j in enumerate(order):
            if i + 2 < len(order):
                q.append(c.submit_batch_host(src[order[i + 2]], stats=(i % 2 == 1)))
            got, st = c.wait_batch(q.pop(0), stats=True)
            for x, y in zip(got, want[j][0]):
                _compare_detect(x, y)
            for k in ("patch_n", "face_patch_n", "cart_gothrough_n", "cart_total_n", "handoff_n"):
                assert st[k] == want[j][1][k], (i, k)
            # the device spans are only measured when Submit was told so (opt->stats as a flag)
            timed = i % 2 == 1
            assert (st["gpu_ms"] > 0) == timed, (i, st["gpu_ms"])
    # single frames (below h2d_min_bytes: the caller's lane) next to a pending ticket
    t = c.submit_batch_host(frames[0])
    one = c.detect(frames[1][0])
    _compare_detect(one, want[1][0][0])
    for x, y in zip(c.wait_batch(t), want[0][0]):
        _compare_detect(x, y)


def test_batch_too_large_for_32bit_window_ids_is_refused(built, gpu, model_file, monkeypatch):
    """Detections carry a 32-bit window id over the whole batch; a batch whose frames x windows exceeds 2^32
    must fail loudly instead of wrapping (the test_wpf_scale option inflates the count the guard sees)."""
    from jda_amd import api, synth
    p, _ = model_file((2, 8, 5, 3), 8, seed=5)
    frames = synth.make_frames(2, 100, 80, seed=6)
    c = api.Cascador(p)
    assert len(c.detect_batch(frames)) == 2
    c.set_option("test_wpf_scale", 1000000000)          # jdaSetOption: the knobs are per cascador, not per call
    assert c.get_option("test_wpf_scale") == 1000000000
    with pytest.raises(api.JdaError, match="batch too large"):
        c.detect_batch(frames)
    c.set_option("test_wpf_scale", 1)
    assert len(c.detect_batch(frames)) == 2
    with pytest.raises(api.JdaError, match="unknown option"):
        c.set_option("no_such_knob", 1)


def test_c_gather_entry_over_rccl_group_of_one(built, gpu, model_file):
    """include/jda_dist.h through libjda_dist.so on its own RCCL communicator (a group of one rank -- all a 1-GPU box
    can host): the exact gather, the pipelined one (empty, full and overflowing blocks -> exact fallback) and the
    gather of jdaResult arrays straight from a detect call."""
    import ctypes as C
    from jda_amd import api, synth, dist as jd
    from jda_amd import build as lib_build
    lib_build.build_dist()
    g = jd.CGather(0, 1, jd.unique_id(), 0, 7, 16)
    m = np.arange(5 * 7, dtype=np.float32).reshape(5, 7)
    assert np.array_equal(g.gather(m), m)
    assert g.gather(np.zeros((0, 7), np.float32)).shape == (0, 7)
    outs = []
    for step in range(4):
        n = [3, 0, 40, 16][step]
        mm = np.full((n, 7), step + 1, np.float32); mm[:, 0] = np.arange(n)
        outs.append(g.start(mm))
    outs.append(g.drain())
    assert outs[0] is None
    for step, got in enumerate(outs[1:]):
        n = [3, 0, 40, 16][step]
        assert got.shape == (n, 7) and (got[:, 1:] == step + 1).all() and list(got[:, 0]) == list(range(n)), (step, got.shape)
    # the first-contact pattern of bench.py --gpus N: two gathers in flight, the first one overflowing its block
    assert jd.gather_selftest(g, 0, 1, 7, 16) == "ok"
    g.close()
    # jdaGatherResults: per-frame results of a detect call -> rows on rank 0 == jdaResultsPack's rows
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-1.0)
    frames = synth.make_frames(3, 200, 150, seed=9)
    c = api.Cascador(p)
    n = len(frames)
    ptrs = (C.POINTER(C.c_ubyte) * n)(*[frames[i].ctypes.data_as(C.POINTER(C.c_ubyte)) for i in range(n)])
    res = (api.jdaResult * n)()
    o = api.jdaDetectOptions(); api.lib.jdaDetectOptionsInit(C.byref(o))
    assert api.lib.jdaDetectBatch(c.h, ptrs, n, 200, 150, 1.25, 0.1, 40, -1, -0.5, C.byref(o), res) == 0
    rows = api.lib.jdaResultsPack(res, n, 100, None, 0)
    want = np.empty((rows, 5 + c.dim), np.float32)
    api.lib.jdaResultsPack(res, n, 100, want.ctypes.data_as(C.POINTER(C.c_float)), rows)
    L = jd.dist_lib()
    L.jdaGatherResults.argtypes = [C.c_void_p, C.POINTER(api.jdaResult), C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int)]
    h = L.jdaDistCreate(0, 1, jd.unique_id(), 0, 5 + c.dim, 64)
    assert h
    out, cnt = C.POINTER(C.c_float)(), C.c_int()
    assert L.jdaGatherResults(h, res, n, 100, C.byref(out), C.byref(cnt)) == 0
    got = np.ctypeslib.as_array(out, (cnt.value, 5 + c.dim)).copy()
    L.jdaDistFree(out); L.jdaDistDestroy(h)
    api.lib.jdaResultsRelease(res, n)
    assert rows > 0 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
