"""BASELINE.json configs[2] (256 x 1920x1080, scale 1.5) under several settings of the JDA_* knobs, one process:
   python tools/config2_variants.py "NAME=VALUE ..." "NAME=VALUE" ...      ("" = defaults)
Frames as in bench.py's configs[2] leg (32 synthesised + 7 cyclic shifts of each, made on the device)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api

f2 = synth.make_frames(32, 1920, 1080, seed=0)
mp = os.path.join(synth.cache_dir(), "config2_5_540_27_4.model")
if not os.path.exists(mp):
    m = synth.make_model(5, 540, 27, 4, seed=1)
    synth.calibrate_thresholds(m, f2[:4], scale=1.5)
    m.save(mp + ".tmp", 8); os.replace(mp + ".tmp", mp)
b2 = torch.from_numpy(f2).cuda()
d = torch.cat([b2] + [torch.roll(b2, shifts=(131 * j, 257 * j), dims=(1, 2)) for j in range(1, 8)]).contiguous()
del b2
steps = int(os.environ.get("VAR_STEPS", "4"))
base_env = dict(os.environ)
ref = None
for spec in (sys.argv[1:] or [""]):
    os.environ.clear(); os.environ.update(base_env)
    for kv in spec.split():
        k, v = kv.split("=", 1); os.environ[k] = v
    c = api.Cascador(mp)
    try:
        for _ in range(2):
            out = c.detect_batch_device(d, 1.5, keep_results="packed")
        digest = (out.shape, float(np.asarray(out, np.float64).sum()))
        if ref is None:
            ref = digest
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sts = []
        for _ in range(steps):
            _, st = c.detect_batch_device(d, 1.5, keep_results=False, stats=True); sts.append(st)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / steps * 1e3
        print("%-50s call %.3f ms  gpu %.3f  scan %.3f  handoff %d  launches %d  dets %d  %s" %
              (spec or "(defaults)", el, np.mean([s["gpu_ms"] for s in sts]), np.mean([s["scan_ms"] for s in sts]),
               sts[-1]["handoff_n"], sts[-1]["scan_launches"], sts[-1]["face_patch_n"], "same" if digest == ref else "DIFFERENT RESULTS"), flush=True)
    except Exception as e:
        print("%-50s FAILED %r" % (spec, e), flush=True)
    c.close()
