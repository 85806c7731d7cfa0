"""k_scan_p, the persistent form of the LDS-tiled stage-0 scan (jda_amd/csrc/k_scan_p.hip; reference loop
c/jda.c:357-402), must give exactly what k_scan gives -- detections bit for bit, the same hand-off count and the same
carts evaluated -- and therefore what the oracle gives.  The option (`scan_p`, JDA_SCAN_P) is read when a cascador is
created: 0 = k_scan only, 2 = k_scan_p wherever it fits (small jobs included, which the default leaves to k_scan)."""
import os

import numpy as np
import pytest

from conftest import S_DIMS, same

pytestmark = pytest.mark.gpu

STAT_KEYS = ("handoff_n", "scan_cart_n", "cart_total_n", "scan_patch_n", "face_patch_n", "patch_n", "cart_gothrough_n")

# bucket boundaries / task forms / ring sizes / workgroup sizes that send the walk through every path of the kernel
VARIANTS = [
    {},
    {"JDA_SCAN_P_B0": "16", "JDA_SCAN_P_B1": "32", "JDA_SCAN_P_B2": "48", "JDA_SCAN_P_B3": "64", "JDA_SCAN_P_B4": "96",
     "JDA_SCAN_P_LG": "66666", "JDA_SCAN_P_BLOCK": "1024"},
    {"JDA_SCAN_P_RING": "64", "JDA_SCAN_P_B2": "96"},                       # small rings: survivors walk on in their task
    {"JDA_SCAN_P_RING": "64", "JDA_SCAN_P_B0": "16", "JDA_SCAN_P_B1": "32", "JDA_SCAN_P_B2": "48", "JDA_SCAN_P_B3": "64",
     "JDA_SCAN_P_B4": "96", "JDA_SCAN_P_LG": "64545", "JDA_SCAN_P_BLOCK": "512"},      # pair tasks of 16 and 32 windows
    {"JDA_SCAN_P_B2": "96", "JDA_SCAN_P_LG": "632", "JDA_SCAN_P_BLOCK": "256", "JDA_SCAN_P_SLOTS": "2"},
    {"JDA_SCAN_P_B0": "8", "JDA_SCAN_P_B1": "24", "JDA_SCAN_P_B2": "40", "JDA_SCAN_P_B3": "100", "JDA_SCAN_P_B4": "120",
     "JDA_SCAN_P_LG": "65454", "JDA_SCAN_P_OPTS": "3", "JDA_SCAN_P_RING": "64"},
    {"JDA_SCAN_P_B0": "4", "JDA_SCAN_P_B1": "0", "JDA_SCAN_P_LG": "4", "JDA_SCAN_P_BLOCK": "128"},
    {"JDA_SCAN_P_B0": "0", "JDA_SCAN_P_B1": "0"},                           # no ring at all: one walk, then the hand-off
    # the kernel's own cut of the tiles in y (more, smaller tiles) and fewer workgroups than CUs
    {"JDA_SCAN_P_TILE_KB": "6", "JDA_SCAN_P_GRID": "24"},
    # all of stage 0 inside the kernel: survivors go straight to the mid queue; deep ranges as pair tasks of 8 / 4 windows
    {"JDA_SCAN_P_HANDOFF": "100000", "JDA_SCAN_P_B2": "128", "JDA_SCAN_P_B3": "256", "JDA_SCAN_P_LG": "64478"},
    {"JDA_SCAN_P_HANDOFF": "100000", "JDA_SCAN_P_TILE_KB": "8", "JDA_SCAN_P_B0": "4", "JDA_SCAN_P_B1": "8", "JDA_SCAN_P_B2": "12",
     "JDA_SCAN_P_B3": "40", "JDA_SCAN_P_LG": "68874", "JDA_SCAN_P_RING": "64"},
    {"JDA_SCAN_P_HANDOFF": "100000", "JDA_SCAN_P_MID": "0", "JDA_SCAN_P_B2": "100", "JDA_SCAN_P_LG": "647"},   # ... through k_filter0
]
# (a variant with its own hand-off evaluates other carts inside the scan and hands other windows over -- and a pass that
# finds most windows alive at ITS hand-off starts over in dense mode, which counts no scan at all)
SCAN_KEYS = ("handoff_n", "scan_cart_n", "scan_patch_n")


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda", 0)


def _run(path, frames_dev, env, th):
    from jda_amd import api
    old = dict(os.environ)
    os.environ.update(env)
    try:
        c = api.Cascador(path)          # (the options are read here)
    finally:
        os.environ.clear(); os.environ.update(old)
    out1, st1 = c.detect_batch_device(frames_dev, th=th, stats=True)
    out2, st2 = c.detect_batch_device(frames_dev, th=th, stats=True)      # second pass: finishing launches sized by prediction
    c.close()
    return out1, st1, out2, st2


def _same_dets(a, b):
    return len(a) == len(b) and all(same(x[k], y[k]) for x, y in zip(a, b) for k in x)


def _check(path, frames, th, variants, oracle_frames=0):
    import torch
    dev = torch.from_numpy(frames).cuda()
    base = {"JDA_MERGE_BLOCKS": "0"}          # a launch per level, so that k_scan_p gets every LDS-tiled level it fits
    ref, st, _, _ = _run(path, dev, dict(base, JDA_SCAN_P="0"), th)
    if oracle_frames:
        from oracle.pyoracle import Oracle
        o = Oracle(path)
        for i in range(oracle_frames):
            want = o.detect(frames[i], th=th)
            assert all(same(want[k], ref[i][k]) for k in want), i
    for v in variants:
        got1, s1, got2, s2 = _run(path, dev, dict(base, JDA_SCAN_P="2", **v), th)
        assert _same_dets(ref, got1) and _same_dets(ref, got2), v
        for k in STAT_KEYS:
            if "JDA_SCAN_P_HANDOFF" in v and k in SCAN_KEYS:
                continue
            assert st[k] == s1[k] == s2[k], (v, k, st[k], s1[k], s2[k])
        assert s1["scan_launches"] > 0


@pytest.mark.parametrize("dims,cart_th,size,n", [
    ((3, 20, 5, 4), -1.0, (200, 150), 3),
    ((3, 20, 5, 4), None, (200, 150), 2),          # nothing is rejected: every window is handed off
    ((2, 8, 5, 3), -0.3, (203, 151), 3),           # odd width: tile loads without LDS-DMA
    ((3, 70, 9, 5), -1.0, (202, 150), 2),          # depth 5, rows aligned to 4 bytes only (and normalising carts)
    ((2, 64, 68, 6), -1.0, (200, 150), 2),
    ((1, 4, 3, 2), -0.3, (200, 150), 2),
])
def test_persistent_scan_equals_k_scan_and_the_oracle_on_small_models(built, gpu, model_file, dims, cart_th, size, n):
    from jda_amd import synth
    p, _ = model_file(dims, 8, seed=3, cart_th=synth.NEG_BIG if cart_th is None else cart_th, norm_every=5)
    frames = synth.make_frames(n, size[0], size[1], seed=11)
    _check(p, frames, -0.5, VARIANTS, oracle_frames=1)


def test_persistent_scan_with_the_shipped_dimensions(built, gpu, model_file):
    from jda_amd import synth
    p, _ = model_file(S_DIMS, 8, seed=3, cart_th=-2.0, norm_every=5)
    _check(p, synth.make_frames(4, 320, 240, seed=11), -0.5, VARIANTS, oracle_frames=1)
    _check(p, synth.make_frames(16, 640, 480, seed=12), -0.5, VARIANTS[:4] + VARIANTS[8:])


def test_persistent_scan_is_what_a_large_batch_runs_by_default(built, gpu, model_file):
    """256 x 640x480 with the cascade-regime model of the bench: the default options send the 46- and 57-pixel levels
    through k_scan_p (scan_p = 1) -- same detections and counters as with it switched off."""
    import torch
    from jda_amd import synth
    import bench
    calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
    mp = bench.model_path(S_DIMS, "cascade", 1, calib)
    dev = torch.from_numpy(synth.make_frames(64, 640, 480, seed=0)).cuda()
    ref, st, _, _ = _run(mp, dev, {"JDA_SCAN_P": "0"}, 0.0)
    got, s1, got2, s2 = _run(mp, dev, {}, 0.0)
    assert _same_dets(ref, got) and _same_dets(ref, got2)
    for k in STAT_KEYS:
        assert st[k] == s1[k] == s2[k], (k, st[k], s1[k], s2[k])


WD_SCRIPT = r"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
from jda_amd import api, synth
assert api.LIB_PATH.endswith("libjda_wd.so"), api.LIB_PATH
frames = torch.from_numpy(synth.make_frames(%(n)d, 640, 480, seed=77)).cuda()
c = api.Cascador(%(model)r)
out = {}
for name, call in (("sync", lambda: c.detect_batch_device(frames, stats=True)),
                   ("ticket", lambda: c.wait_batch(c.submit_batch_device(frames), stats=True))):
    dets, st = call()
    err = api.last_error()
    out[name] = {"err": err, "fallbacks": st["scan_fallbacks"], "scan_patch_n": st["scan_patch_n"], "patch_n": st["patch_n"], "cart_total_n": st["cart_total_n"],
                 "n": [len(d["scores"]) for d in dets],
                 "digest": [__import__("hashlib").sha256(b"".join(np.ascontiguousarray(d[k]).tobytes() for k in ("bboxes", "scores", "shapes"))).hexdigest() for d in dets]}
# a ragged job too (the persistent kernel's RAGGED instantiation: tiles from the chunk's block map)
rng = np.random.default_rng(3)
base = synth.make_frames(8, 400, 300, seed=79)
imgs = [np.ascontiguousarray(base[i %% 8][:int(rng.integers(200, 301)), :int(rng.integers(260, 401))]) for i in range(%(n_rag)d)]
dets, st = c.detect_ragged(imgs, stats=True)
out["ragged"] = {"err": api.last_error(), "fallbacks": st["scan_fallbacks"], "n": [len(d["scores"]) for d in dets],
                 "digest": [__import__("hashlib").sha256(b"".join(np.ascontiguousarray(d[k]).tobytes() for k in ("bboxes", "scores", "shapes"))).hexdigest() for d in dets]}
print("RESULT " + json.dumps(out))
"""


def test_a_tripped_watchdog_is_noticed_and_the_pass_rerun(gpu, tmp_path):
    """k_scan_p's wait loops have watchdogs (a scheduling bug must not hang the device).  A launch that trips one loses
    windows; it used to return them short with rc 0.  Now the kernel sets an error word, the host also compares the
    windows covered with the plan's, and either way runs the pass again with k_scan's closed tiles: correct results,
    a note on stderr and jdaStats.scan_fallbacks -- NOT in jdaGetLastError(): an empty result with a non-empty error
    string is how jdaDetect reports failure, and a recovered pass that finds no face is not one.  libjda_wd.so (jda_amd/build.py:build_watchdog) is the product with the
    idle watchdog of k_scan_p.hip at zero -- every persistent launch trips it."""
    import hashlib
    import json
    import subprocess
    import sys
    import torch
    from jda_amd import api, synth, build as lib_build
    wd = lib_build.build_watchdog()
    n = 64
    path = os.path.join(synth.cache_dir(), "wd_%d_%d_%d_%d.model" % S_DIMS)
    if not os.path.exists(path):
        m = synth.make_model(*S_DIMS, seed=5)
        synth.calibrate_thresholds(m, synth.make_frames(2, 640, 480, seed=78))
        m.save(path + ".tmp", 8); os.replace(path + ".tmp", path)
    frames = torch.from_numpy(synth.make_frames(n, 640, 480, seed=77)).to(gpu)
    old = dict(os.environ); os.environ["JDA_SCAN_P"] = "0"
    try:
        c = api.Cascador(path)
    finally:
        os.environ.clear(); os.environ.update(old)
    want, st = c.detect_batch_device(frames, stats=True)
    n_rag = 400
    rng = np.random.default_rng(3)
    base = synth.make_frames(8, 400, 300, seed=79)
    imgs = [np.ascontiguousarray(base[i % 8][:int(rng.integers(200, 301)), :int(rng.integers(260, 401))]) for i in range(n_rag)]
    want_rag = c.detect_ragged(imgs)
    c.close()
    assert sum(len(d["scores"]) for d in want) > 0 and sum(len(d["scores"]) for d in want_rag) > 0
    env = dict(os.environ, JDA_LIB_PATH=wd, JDA_SCAN_P="2")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", WD_SCRIPT % dict(root=root, n=n, model=path, n_rag=n_rag)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert "run again with k_scan" in r.stderr, r.stderr[-2000:]
    digest = [hashlib.sha256(b"".join(np.ascontiguousarray(d[k]).tobytes() for k in ("bboxes", "scores", "shapes"))).hexdigest() for d in want]
    for name in ("sync", "ticket"):
        g = got[name]
        assert g["err"] == "" and g["fallbacks"] >= 1, (name, g["err"], g["fallbacks"])
        assert g["n"] == [len(d["scores"]) for d in want] and g["digest"] == digest, name
        assert g["scan_patch_n"] == st["scan_patch_n"] == g["patch_n"] and g["cart_total_n"] == st["cart_total_n"], name
    g = got["ragged"]
    assert g["err"] == "" and g["fallbacks"] >= 1, (g["err"], g["fallbacks"])
    assert g["n"] == [len(d["scores"]) for d in want_rag]
    assert g["digest"] == [hashlib.sha256(b"".join(np.ascontiguousarray(d[k]).tobytes() for k in ("bboxes", "scores", "shapes"))).hexdigest() for d in want_rag]
