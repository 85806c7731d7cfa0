"""PCIe-inclusive step of BASELINE.json configs[1] (frames in host memory, jdaDetectBatchSubmitHost / Wait, two batches
in flight) under several settings:  python tools/host_variants.py "NAME=V NAME=V" ...   (pageable and pinned sources)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jda_amd import synth, api
import bench
dims = (5, 540, 27, 4)
mp = bench.model_path(dims, "cascade", 1, synth.make_frames(8, 640, 480, seed=0, first=10_000_000))
src = [synth.make_frames(256, 640, 480, seed=0, first=j * 256) for j in range(2)]
pin = [torch.from_numpy(f).pin_memory() for f in src]
wpf = api.count_windows(640, 480, 1.25, 40, -1)[0]
base = dict(os.environ)
for spec in (sys.argv[1:] or [""]):
    os.environ.clear(); os.environ.update(base)
    for kv in spec.split():
        k, v = kv.split("=", 1); os.environ[k] = v
    out = []
    for name, frames in (("pageable", src), ("pinned", [p.numpy() for p in pin])):
        if os.environ.get("HOST_ONLY", name) != name:
            continue
        import _dummy_streams; _dummy_streams.make()
        c = api.Cascador(mp)
        ahead = int(os.environ.get("AHEAD", "2"))
        def run(steps):
            q = [c.submit_batch_host(frames[j % 2]) for j in range(min(ahead, steps))]
            for i in range(steps):
                if i + ahead < steps:
                    q.append(c.submit_batch_host(frames[(i + ahead) % 2]))
                c.wait_batch(q.pop(0), keep_results="packed")
        run(4)
        torch.cuda.synchronize(); t0 = time.perf_counter(); steps = int(os.environ.get('HOST_STEPS', '30'))
        run(steps)
        el = (time.perf_counter() - t0) / steps
        out.append("%s %.3f ms %.2e win/s" % (name, el * 1e3, wpf * 256 / el))
        c.close()
    print("%-50s %s" % (spec or "(defaults)", "   ".join(out)), flush=True)
