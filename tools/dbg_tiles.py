import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, synth.make_frames(8, 640, 480, seed=0, first=10_000_000)); m.save(mp, 8)
c = api.Cascador(mp)
frames = synth.make_frames(8, 640, 480, seed=0)
d = torch.from_numpy(frames).cuda()
print("== uniform", file=sys.stderr); c.detect_batch_device(d, keep_results=False)
print("== ragged", file=sys.stderr); c.detect_ragged_packed(d.view(-1), [i*640*480 for i in range(8)], [640]*8, [480]*8, keep_results=False)
