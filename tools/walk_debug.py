"""First window whose trace differs between the k_walk path and the old k_finish path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jda_amd import synth, api
dims = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3,20,5,4").split(","))
th = float(sys.argv[2]) if len(sys.argv) > 2 else -1.0
p = "/tmp/walkdbg.model"
synth.make_model(*dims, seed=3, cart_th=th, norm_every=5).save(p, 8)
frames = synth.make_frames(2, 200, 150, seed=11)
res = {}
for walk in ("1", "0"):
    os.environ["JDA_WALK"] = walk
    c = api.Cascador(p)
    res[walk] = c.trace(frames)
    res[walk + "d"] = c.detect_batch(frames, nms=False)
    c.close()
a, b = res["1"], res["0"]
for k in ("carts_n", "score", "path_hash", "shapes"):
    va, vb = a[k], b[k]
    bad = np.nonzero((va.view(np.uint32) if va.dtype == np.float32 else va).reshape(len(a["carts_n"]), -1) !=
                     (vb.view(np.uint32) if vb.dtype == np.float32 else vb).reshape(len(a["carts_n"]), -1))[0]
    print(k, "differs at", len(set(bad.tolist())), "windows of", len(a["carts_n"]))
bad = np.nonzero(a["carts_n"] != b["carts_n"])[0]
K = dims[1]
for i in bad[:10]:
    print("window", i, "walk carts", a["carts_n"][i], "(stage %d cart %d)" % (a["carts_n"][i] // K, a["carts_n"][i] % K), "old carts", b["carts_n"][i],
          "scores", a["score"][i], b["score"][i])
bs = np.nonzero((a["shapes"].view(np.uint32) != b["shapes"].view(np.uint32)).any(1) & (a["carts_n"] == b["carts_n"]))[0]
for i in bs[:5]:
    print("shape-only diff window", i, "carts", a["carts_n"][i], a["shapes"][i][:4], b["shapes"][i][:4])
print("detections walk/old:", [len(d["scores"]) for d in res["1d"]], [len(d["scores"]) for d in res["0d"]])
