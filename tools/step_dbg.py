"""Event/host deltas of a few 256-frame steps (JDA_DEBUG_TIMES=1 prints them on stderr)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jda_amd import synth, api
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, calib); m.save(mp, 8)
c = api.Cascador(mp)
d = torch.from_numpy(synth.make_frames(256, 640, 480, seed=0)).cuda()
for i in range(3): c.detect_batch_device(d, keep_results="packed")
os.environ["JDA_DEBUG_TIMES"] = "1"
for i in range(3):
    t0 = time.perf_counter()
    _, st = c.detect_batch_device(d, keep_results="packed", stats=True)
    print("step %.3f ms call %.3f gpu %.3f host_post %.3f" % ((time.perf_counter() - t0) * 1e3, st["call_ms"], st["gpu_ms"], st["host_ms"]), file=sys.stderr)
