"""BASELINE.json configs[2] at its stated size: 256 x 1920x1080 frames resident in HBM, scale 1.5 (8 window sizes: 40, 60, 90,
135, 202, 303, 455, 683 px; c/jda.c:331-333), shipped model dimensions in the cascade regime -- 32,089,600 windows per call.
   python tools/config2.py [sync|pipe] [steps]     (run it under rocprofv3 for the kernel trace / counter passes)
sync: one jdaDetectBatchDevice call per step; pipe: jdaDetectBatchSubmit / Wait, CFG2_AHEAD (2) batches queued ahead (the host's
sort + NMS of one batch behind the others' kernels).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api

mode = sys.argv[1] if len(sys.argv) > 1 else "sync"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(os.environ.get("CFG2_BATCH", "256"))
frames = synth.make_frames(n, 1920, 1080, seed=0)
mp = os.path.join(synth.cache_dir(), "config2_5_540_27_4.model")
if not os.path.exists(mp):
    m = synth.make_model(5, 540, 27, 4, seed=1)
    synth.calibrate_thresholds(m, frames[:4], scale=1.5)
    m.save(mp + ".tmp", 8); os.replace(mp + ".tmp", mp)
c = api.Cascador(mp)
d = torch.from_numpy(frames).cuda()
kw = dict(scale=1.5)
ahead = int(os.environ.get("CFG2_AHEAD", "2"))      # pipe: batches queued ahead of the one being collected (three tickets exist)
def run(steps, sts):
    if mode == "pipe":
        q = [c.submit_batch_device(d, stats=True, **kw) for _ in range(min(ahead, steps))]
        issued = len(q)
        for _ in range(steps):
            if issued < steps:
                q.append(c.submit_batch_device(d, stats=True, **kw)); issued += 1
            _, st = c.wait_batch(q.pop(0), stats=True, keep_results=False); sts.append(st)
    else:
        for _ in range(steps):
            _, st = c.detect_batch_device(d, keep_results=False, stats=True, **kw); sts.append(st)
run(4, [])              # warm-up in the mode that is timed: a ticket's lane holds the whole batch (its workspace is allocated here)
torch.cuda.synchronize()
sts = []
t0 = time.perf_counter()
run(steps, sts)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / steps
st = sts[-1]
print(json.dumps({"config": "BASELINE.json configs[2]: %d x 1920x1080, scale 1.5 (8 levels), S dims, cascade regime" % n, "mode": mode,
                  "windows": st["patch_n"], "ms_per_step": el * 1e3, "windows_per_s": st["patch_n"] / el,
                  "gpu_ms": float(np.mean([s["gpu_ms"] for s in sts])), "scan_ms": float(np.mean([s["scan_ms"] for s in sts])),
                  "host_ms": float(np.mean([s["host_ms"] for s in sts])), "average_cart_n": st["average_cart_n"],
                  "handoff_n": st["handoff_n"], "detections": st["face_patch_n"], "scan_launches": st["scan_launches"]}))
