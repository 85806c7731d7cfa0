"""Cascade-regime step time of BASELINE.json configs[1] under several settings of the JDA_* experiment knobs.
   python tools/variants.py "NAME=VALUE NAME=VALUE" "NAME=VALUE" ...      (one argument per variant; "" = defaults)
Every variant gets a fresh cascador (tile plans are cached per cascador).  Prints ms per synchronous step, the device
span, the k_scan span (HIP events) and the hand-off count; JDA_LANES=1 JDA_SIDE_STREAM=0 serialises the launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dims = (5, 540, 27, 4)
calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
mp = bench.model_path(dims, "cascade", 1, calib)
B = int(os.environ.get("VAR_BATCH", "256"))
d = torch.from_numpy(synth.make_frames(B, 640, 480, seed=0)).cuda()
steps = int(os.environ.get("VAR_STEPS", "10"))
base_env = dict(os.environ)
for spec in (sys.argv[1:] or [""]):
    os.environ.clear(); os.environ.update(base_env)
    for kv in spec.split():
        k, v = kv.split("=", 1); os.environ[k] = v
    c = api.Cascador(mp)
    try:
        for _ in range(3):
            c.detect_batch_device(d, keep_results=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sts = []
        for _ in range(steps):
            _, st = c.detect_batch_device(d, keep_results=False, stats=True); sts.append(st)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / steps * 1e3
        g = np.mean([s["gpu_ms"] for s in sts]); sc = np.mean([s["scan_ms"] for s in sts])
        lds = np.mean([s["scan_lds_ms"] for s in sts])
        print("%-60s step %.3f ms  gpu %.3f  scan %.3f  lds %.3f  handoff %d  scan_carts %d  launches %d  dets %d" %
              (spec or "(defaults)", el, g, sc, lds, sts[-1]["handoff_n"], sts[-1]["scan_cart_n"], sts[-1]["scan_launches"], sts[-1]["face_patch_n"]), flush=True)
    except Exception as e:
        print("%-60s FAILED %r" % (spec, e), flush=True)
    c.close()
