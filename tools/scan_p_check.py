"""k_scan_p (persistent scan) against k_scan on the same inputs: detections bit for bit, hand-off count, carts evaluated.
   python tools/scan_p_check.py [quick]
Each case runs with JDA_SCAN_P=0 and JDA_SCAN_P=2 (a fresh cascador each; the option is read at creation) under several
bucket / block settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api

def bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a

def run(path, frames_dev, env, th):
    old = dict(os.environ)
    os.environ.update(env)
    c = api.Cascador(path)
    os.environ.clear(); os.environ.update(old)
    out, st = c.detect_batch_device(frames_dev, th=th, stats=True)
    out2, st2 = c.detect_batch_device(frames_dev, th=th, stats=True)      # second pass: predicted path
    c.close()
    return out, st, out2, st2

def same(a, b):
    if len(a) != len(b): return False
    for x, y in zip(a, b):
        for k in x:
            if x[k].shape != y[k].shape or not np.array_equal(bits(x[k]), bits(y[k])): return False
    return True

VARIANTS = [
    {"JDA_SCAN_P": "2"},
    {"JDA_SCAN_P": "2", "JDA_SCAN_P_B0": "16", "JDA_SCAN_P_B1": "32", "JDA_SCAN_P_B2": "48", "JDA_SCAN_P_B3": "64", "JDA_SCAN_P_B4": "96", "JDA_SCAN_P_LG": "66666", "JDA_SCAN_P_BLOCK": "1024"},
    {"JDA_SCAN_P": "2", "JDA_SCAN_P_RING": "64", "JDA_SCAN_P_B2": "96"},                     # small rings: survivors walk on in their task
    {"JDA_SCAN_P": "2", "JDA_SCAN_P_RING": "64", "JDA_SCAN_P_B0": "16", "JDA_SCAN_P_B1": "32", "JDA_SCAN_P_B2": "48", "JDA_SCAN_P_B3": "64", "JDA_SCAN_P_B4": "96", "JDA_SCAN_P_LG": "64545", "JDA_SCAN_P_BLOCK": "512"},
    {"JDA_SCAN_P": "2", "JDA_SCAN_P_B2": "96", "JDA_SCAN_P_LG": "632", "JDA_SCAN_P_BLOCK": "256", "JDA_SCAN_P_SLOTS": "2"},
    {"JDA_SCAN_P": "2", "JDA_SCAN_P_B0": "8", "JDA_SCAN_P_B1": "24", "JDA_SCAN_P_B2": "40", "JDA_SCAN_P_B3": "100", "JDA_SCAN_P_B4": "120", "JDA_SCAN_P_LG": "65454", "JDA_SCAN_P_OPTS": "3", "JDA_SCAN_P_RING": "64"},
    {"JDA_SCAN_P": "2", "JDA_SCAN_P_B0": "4", "JDA_SCAN_P_B1": "0", "JDA_SCAN_P_LG": "4", "JDA_SCAN_P_BLOCK": "128"},
    {"JDA_SCAN_P": "2", "JDA_SCAN_P_B0": "0", "JDA_SCAN_P_B1": "0"},
]

def case(dims, cart_th, size, n, th=-0.5, seed=3, norm_every=5, variants=VARIANTS):
    m = synth.make_model(*dims, seed=seed, cart_th=cart_th, norm_every=norm_every)
    p = "/tmp/sp_%d_%d_%d_%d_%d.model" % (dims + (seed,))
    m.save(p, 8)
    fr = torch.from_numpy(synth.make_frames(n, size[0], size[1], seed=11)).cuda()
    base_env = {"JDA_SCAN_P": "0", "JDA_MERGE_BLOCKS": "0"}         # per-level launches, so that k_scan_p gets every LDS-tiled level
    ref, st, ref2, _ = run(p, fr, base_env, th)
    bad = 0
    for v in variants:
        env = dict(base_env); env.update(v)
        t0 = time.time()
        got, st1, got2, st2 = run(p, fr, env, th)
        ok = same(ref, got) and same(ref, got2)
        keys = ("handoff_n", "scan_cart_n", "cart_total_n", "scan_patch_n", "face_patch_n", "patch_n", "cart_gothrough_n")
        stat_ok = all(st.get(k) == st1.get(k) == st2.get(k) for k in keys if k in st)
        if not (ok and stat_ok): bad += 1
        print("%-22s th %-8g %dx%d x%d  %-110s dets %s stats %s  %s  (%.1fs)" % (
            dims, cart_th, size[0], size[1], n, " ".join("%s=%s" % (k[4:], x) for k, x in v.items()),
            "ok" if ok else "MISMATCH", "ok" if stat_ok else "MISMATCH %r vs %r" % ({k: st.get(k) for k in keys}, {k: st1.get(k) for k in keys}),
            sum(len(d["scores"]) for d in ref), time.time() - t0), flush=True)
    return bad

if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    bad = 0
    bad += case((3, 20, 5, 4), -1.0, (200, 150), 3)
    bad += case((5, 540, 27, 4), -2.0, (320, 240), 4)
    if not quick:
        bad += case((3, 20, 5, 4), synth.NEG_BIG, (200, 150), 2)          # nothing is rejected: every window is handed off
        bad += case((2, 8, 5, 3), -0.3, (203, 151), 3)                      # odd width: tile loads without LDS-DMA
        bad += case((3, 70, 9, 5), -1.0, (202, 150), 2)                     # depth 5, 4-byte aligned rows
        bad += case((2, 64, 68, 6), -1.0, (200, 150), 2)
        bad += case((1, 4, 3, 2), -0.3, (200, 150), 2)
        bad += case((5, 540, 27, 4), -2.0, (640, 480), 16, variants=VARIANTS[:4])
        bad += case((5, 540, 27, 4), synth.NEG_BIG, (160, 120), 1, variants=VARIANTS[:2])
    print("TOTAL BAD", bad)
    sys.exit(1 if bad else 0)
