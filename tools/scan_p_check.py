"""k_scan_p (persistent scan) against k_scan on the same inputs: detections bit for bit, hand-off count, carts evaluated --
the cases of tests/test_scan_persistent.py as a script that prints a line per variant (quick: two cases).
   python tools/scan_p_check.py [quick]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from jda_amd import synth
import test_scan_persistent as T

def case(dims, cart_th, size, n, th=-0.5, variants=T.VARIANTS):
    m = synth.make_model(*dims, seed=3, cart_th=cart_th, norm_every=5)
    p = "/tmp/sp_%d_%d_%d_%d.model" % dims
    m.save(p, 8)
    dev = torch.from_numpy(synth.make_frames(n, size[0], size[1], seed=11)).cuda()
    base = {"JDA_MERGE_BLOCKS": "0"}
    ref, st, _, _ = T._run(p, dev, dict(base, JDA_SCAN_P="0"), th)
    bad = 0
    for v in variants:
        t0 = time.time()
        g1, s1, g2, s2 = T._run(p, dev, dict(base, JDA_SCAN_P="2", **v), th)
        ok = T._same_dets(ref, g1) and T._same_dets(ref, g2)
        sok = all(st[k] == s1[k] == s2[k] for k in T.STAT_KEYS if not ("JDA_SCAN_P_HANDOFF" in v and k in T.SCAN_KEYS))
        bad += not (ok and sok)
        if not sok:
            print("   stats differ:", {k: (st[k], s1[k], s2[k]) for k in T.STAT_KEYS if not (st[k] == s1[k] == s2[k])}, flush=True)
        print("%-22s th %-8g %dx%d x%d  %-120s dets %s stats %s  %d  (%.1fs)" % (dims, cart_th, size[0], size[1], n,
              " ".join("%s=%s" % (k[4:], x) for k, x in v.items()), "ok" if ok else "MISMATCH", "ok" if sok else "MISMATCH",
              sum(len(d["scores"]) for d in ref), time.time() - t0), flush=True)
    return bad

if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    if len(sys.argv) > 1 and sys.argv[1] == "none": sys.exit(0)
    bad = case((3, 20, 5, 4), -1.0, (200, 150), 3) + case((5, 540, 27, 4), -2.0, (320, 240), 4)
    if not quick:
        bad += case((3, 20, 5, 4), synth.NEG_BIG, (200, 150), 2) + case((2, 8, 5, 3), -0.3, (203, 151), 3)
        bad += case((3, 70, 9, 5), -1.0, (202, 150), 2) + case((2, 64, 68, 6), -1.0, (200, 150), 2) + case((1, 4, 3, 2), -0.3, (200, 150), 2)
        bad += case((5, 540, 27, 4), -2.0, (640, 480), 16, variants=T.VARIANTS[:4] + T.VARIANTS[8:])
    print("TOTAL BAD", bad)
    sys.exit(1 if bad else 0)
