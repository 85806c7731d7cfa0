#!/usr/bin/env python3
"""BASELINE.json configs[3]: an FDDB-shaped job (2,845 images <= 450x450, 10 folds) through the
FDDB harness (jda_amd/fddb.py), sharded over the ranks of torch.distributed, detections gathered
on rank 0 which writes the ten fold-XX-out.txt files.  Reports images/s of the detect loop
(images decoded beforehand, like the reference times only Detect, src/test.cpp:140-144).

  python tools/fddb_bench.py [--images 2845] [--dialect cpp|c] [--threads 4]
  python -m torch.distributed.run --nproc-per-node N tools/fddb_bench.py ..."""
import argparse, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2845)
    ap.add_argument("--dialect", default="cpp")
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--dir", default="/tmp/jda_fddb_synth")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from jda_amd import api, fddb, synth, dist as jdist
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    d = "%s_%d" % (args.dir, args.images)
    if rank == 0 and not os.path.exists(os.path.join(d, "FDDB-folds", "FDDB-fold-10.txt")):
        t0 = time.time(); fddb.make_synthetic_fddb(d, args.images, seed=0); print("dataset s %.1f" % (time.time() - t0), file=sys.stderr)
    if world > 1:
        dist.barrier()
    # cascade-regime model with the shipped dimensions, calibrated on a few of the images' sizes
    mp = os.path.join(synth.cache_dir(), "fddb_model_%s.model" % args.dialect)
    if not os.path.exists(mp):
        m = synth.make_model(5, 540, 27, 4, seed=1)
        synth.calibrate_thresholds(m, synth.make_frames(8, 450, 450, seed=0, first=10_000_000), scale=1.25, min_size=40)
        m.save(mp + ".%d" % os.getpid(), 8); os.replace(mp + ".%d" % os.getpid(), mp)
    job = fddb.list_job(d)
    lo, hi = jdist.shard_range(len(job), rank, world)
    t0 = time.time()
    grays = [fddb.load_gray(os.path.join(d, "images", job[i][1] + ".jpg")) for i in range(lo, hi)]
    decode_s = time.time() - t0
    casc = api.Cascador(mp, device=local)           # ONE cascador for every host thread: jdaDetect is re-entrant
    cascs = [casc] * args.threads
    def work(t):
        out = []
        for j in range(t, len(grays), args.threads):
            out.append((j, fddb.detect_image(cascs[t], grays[j], args.dialect)))
        return out
    work(0)[:0]
    for t in range(args.threads):                       # warm-up: plans for a few sizes, model upload
        fddb.detect_image(cascs[t], grays[t % len(grays)], args.dialect)
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(args.threads) as ex:
        parts = list(ex.map(work, range(args.threads)))
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); el = float(t.item())
    windows = sum(st["patch_n"] for p in parts for _, (_, _, _, st) in p)
    ndet = sum(len(sc) for p in parts for _, (_, sc, _, _) in p)
    if world > 1:
        t = torch.tensor([windows, ndet], dtype=torch.float64, device=dev); dist.all_reduce(t); windows, ndet = int(t[0]), int(t[1])
    # the full harness (decode + detect + gather + files), untimed part of the record
    t1 = time.perf_counter()
    fddb.run(cascs[0], d, dialect=args.dialect, rank=rank, world=world, device=dev)
    full_s = time.perf_counter() - t1
    if rank == 0:
        print(json.dumps({"metric": "FDDB-shaped images/sec (detect loop, images decoded beforehand)",
                          "value": len(job) / el, "unit": "images/s", "n_gpus": world, "images": len(job),
                          "windows_per_s": windows / el, "detections": ndet, "dialect": args.dialect,
                          "host_threads_per_gpu": args.threads, "decode_s_per_rank": decode_s,
                          "harness_end_to_end_s": full_s, "data": "synthetic FDDB layout, sizes <= 450x450"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
