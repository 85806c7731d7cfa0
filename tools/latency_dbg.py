"""Event deltas of single-frame calls (JDA_DEBUG_TIMES=1 prints them on stderr)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["JDA_DEBUG_TIMES"] = "0"
from jda_amd import synth, api
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, calib); m.save(mp, 8)
c = api.Cascador(mp)
f = synth.make_frames(4, 640, 480, seed=1)
for i in range(20): c.detect(f[i % 4])
os.environ["JDA_DEBUG_TIMES"] = "1"
for i in range(6): c.detect(f[i % 4])
