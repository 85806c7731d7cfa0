#!/usr/bin/env python3
"""Throughput benchmark of the jdaDetect hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the detect path (jdaDetectBatchDevice: stage-0 scan,
later stages, regression, compaction, D2H of survivors, host NMS+relocation)
over one batch of synthetic frames that is already resident in HBM.  The steps
cycle through --rotate (4) distinct resident batches (314 MB of frames), so a
step does not find its frames in the 256 MB Infinity Cache from the step before.
The PCIe-inclusive rate (frames in host memory, jdaDetectBatchSubmitHost) is
reported in config.host_frames_windows_per_s and is never `value`.  At N>1
the driver launches one process per GPU with torch.distributed.run; every rank
runs its own batch (weak scaling, no data-path collective) and the detections
are gathered on rank 0 over RCCL inside the timed region (the gather of step i
overlaps the detection of step i+1; the last one is drained before the clock stops).

Workload = BASELINE.json configs[1]: batch of 256 frames 640x480, synthetic
model with the shipped dimensions (T=5, K=540, 27 landmarks, depth 4), canonical
call jdaDetect(.., 1.25, 0.1, 40, -1, -0.5) (reference c/main.cpp:25), in the
"cascade" threshold regime (mean reject length ~30 carts, ~0.1 % of windows
finish).  The all-pass regime is measured as a secondary line in "regimes".

The K timed steps go through the library's submit/wait entry points (jdaDetectBatchSubmit /
jdaDetectBatchWait): step i+1 is queued before step i is collected (--depth 2; --depth 3 queues two ahead, the library
has three tickets), so two batches are in flight per GPU from ONE host thread and the GPU works while the host parts of a step
run (r06, on streams that each have a hardware queue: two in flight 1.34-1.36 ms per step in steady state, three 1.36-1.37; over 20
timed steps from an empty pipeline 1.37-1.39 against 1.40-1.41 -- three passes started at once ramp up slower).  Every step is a complete
pass over one batch and all K finish inside the timed region.  --depth 1 uses one synchronous
jdaDetectBatchDevice call per step instead; that figure is reported next to the headline in "config" and
"regimes".

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# LDS: 64 banks x 4 B per CU and clock, but a 4-byte-or-narrower read is served 32 lanes per clock (128 B/clk/CU);
# 256 CUs at 2.4 GHz (same guide, section LDS: "~75 TB/s for ds_read_b32")
LDS_PEAK_GBPS = 128 * 256 * 2.4
# what 64 lanes reading random bytes reach on this part: 8.2 LDS clocks per wave instruction instead of 2
# (measured, tools/lds_bench.hip -> profiles/r02_lds_bench.txt)
LDS_RANDOM_GATHER_GBPS = LDS_PEAK_GBPS * 2.0 / 8.2


def algorithmic_bytes(dims, carts, stage_done, windows, accepted):
    """B = c*[(D-1)*34+16] + s*K*2L*4 + 2L*4 per window (+16+8L per accepted)."""
    T, K, L, D = dims
    per_cart = (D - 1) * 34 + 16
    per_stage = K * 2 * L * 4
    return carts * per_cart + sum(stage_done) * per_stage + windows * 2 * L * 4 + accepted * (16 + 8 * L)


def model_path(dims, regime, seed, calib_frames):
    from jda_amd import synth
    tag = "%d_%d_%d_%d_%s_s%d" % (dims + (regime, seed))
    path = os.path.join(synth.cache_dir(), "model_%s.model" % tag)
    if os.path.exists(path):
        return path
    m = synth.make_model(*dims, seed=seed)
    if regime == "cascade":
        synth.calibrate_thresholds(m, calib_frames)
    tmp = path + ".%d.tmp" % os.getpid()
    m.save(tmp, 8)
    os.replace(tmp, path)
    return path


def cpu_baseline(model_file, dims, frames, budget_s=20.0):
    """Reference CPU path on the host cores, bounded sample of the same frames/model/call."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle
    from jda_amd import synth
    kind, runner = None, None
    try:
        ref = pyoracle.Reference(model_file, dims, 8)
        kind = "reference"
        runner = lambda img: ref.detect(img, 1.25, 40, -1, -0.5)
    except Exception:
        orc = pyoracle.Oracle(model_file)
        kind = "port"
        runner = lambda img: orc.detect(img, 1.25, 40, -1, -0.5)
    h, w = frames.shape[1:]
    wpf = synth.levels_c(w, h)[1]
    # single thread: the reference as shipped (no parallel region in c/jda.c)
    t0 = time.perf_counter(); runner(frames[0]); one = time.perf_counter() - t0
    n1 = int(max(2, min(len(frames), (budget_s * 0.3) / max(one, 1e-4))))
    t0 = time.perf_counter()
    for i in range(n1):
        runner(frames[i])
    t1 = time.perf_counter() - t0
    single = n1 * wpf / t1
    # image-parallel over all host cores: the only parallel form the reference
    # itself uses for detection (src/test.cpp:100)
    cores = os.cpu_count() or 1
    n2 = int(max(cores, min(len(frames), (budget_s * 0.7) / max(one, 1e-4) * cores)))
    n2 = min(n2, len(frames))
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(runner, [frames[i] for i in range(n2)]))
        t2 = time.perf_counter() - t0
    multi = n2 * wpf / t2
    return {"value": multi, "unit": "windows/s", "cores": cores, "kind": kind,
            "sample": "%d of the batch's frames, one jdaDetect per host thread (%d threads); "
                      "NMS+relocation included" % (n2, cores),
            "single_thread_value": single, "single_thread_sample": "%d frames" % n1}


def make_checker(model_file, dims):
    """The CPU answer for one image: the compiled reference c/jda.c (oracle/_ref, kind "reference") where a build for these
    dimensions exists, else the oracle's restatement ("port").  Test infrastructure, used AFTER the timed regions only."""
    from oracle import pyoracle
    try:
        ref = pyoracle.Reference(model_file, dims, 8)
        return "reference", ref.detect
    except Exception:
        orc = pyoracle.Oracle(model_file)
        return "port", orc.detect


def rows_equal(rows, index, want):
    """Are the packed rows [index, x, y, size, score, shape...] of image `index` exactly the detections `want`
    (dict bboxes / scores / shapes of the checker)?  Bit for bit (float32 viewed as uint32)."""
    rows = np.asarray(rows, np.float32)
    got = rows[rows[:, 0] == np.float32(index)] if len(rows) else rows.reshape(0, rows.shape[1] if rows.ndim == 2 else 0)
    n = len(want["scores"])
    if len(got) != n:
        return False
    if n == 0:
        return True
    exp = np.concatenate([np.full((n, 1), index, np.float32), want["bboxes"].astype(np.float32),
                          want["scores"].reshape(n, 1).astype(np.float32), want["shapes"].astype(np.float32)], axis=1)
    return bool(np.array_equal(np.ascontiguousarray(got).view(np.uint32), np.ascontiguousarray(exp).view(np.uint32)))


def rows_equal_d(rows, index, want):
    """The same for dialect CPP's float64 rows [index, x, y, w, h, score, shape...] against dict rects / scores / shapes."""
    rows = np.asarray(rows, np.float64)
    got = rows[rows[:, 0] == float(index)] if len(rows) else rows.reshape(0, rows.shape[1] if rows.ndim == 2 else 0)
    n = len(want["scores"])
    if len(got) != n:
        return False
    if n == 0:
        return True
    exp = np.concatenate([np.full((n, 1), float(index)), want["rects"].astype(np.float64),
                          want["scores"].reshape(n, 1), want["shapes"]], axis=1)
    return bool(np.array_equal(np.ascontiguousarray(got).view(np.uint64), np.ascontiguousarray(exp).view(np.uint64)))


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_spawn_argv(argv, gpus, port):
    """The command `python bench.py --gpus N ...` re-executes itself as when N > 1 and it was not started by a
    launcher (no RANK / WORLD_SIZE in the environment): one process per GPU, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def maybe_self_spawn(args, argv):
    """--gpus N > 1 without a launcher: become the launcher.  Returns the exit code of the N-rank job, or None when
    this process is itself a rank (or N == 1) and should go on."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return None
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and os.environ.get("JDA_BENCH_ONE_GPU") != "1":
        sys.stderr.write("bench.py: --gpus %d asked for, %d HIP device(s) visible on this node\n" % (args.gpus, have))
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(self_spawn_argv(argv, args.gpus, free_port()), env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of this node; N > 1 without a launcher re-executes under torch.distributed.run")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--rotate", type=int, default=4,
                    help="distinct device-resident batches the steps cycle through (4 x 78.6 MB > the 256 MB Infinity Cache)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--dims", type=str, default="5,540,27,4")
    ap.add_argument("--depth", type=int, default=2,
                    help="batches in flight per rank: 2 / 3 = submit/wait pipeline from one host thread, one / two batches queued ahead (default 2), "
                         "1 = one synchronous call per step")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-allpass", action="store_true")
    ap.add_argument("--no-x", action="store_true", help="skip the configs[4] all-pass leg (roofline_hbm_regime)")
    ap.add_argument("--no-config2", action="store_true",
                    help="skip the BASELINE.json configs[2] leg (256 x 1080p, scale 1.5: config.config2_*)")
    ap.add_argument("--config2", action="store_true", help="(accepted for older command lines: the leg runs by default)")
    args = ap.parse_args()
    rc = maybe_self_spawn(args, sys.argv[1:])
    if rc is not None:
        raise SystemExit(rc)

    import torch
    import torch.distributed as dist
    if not os.path.exists(os.path.join(ROOT, "jda_amd", "libjda.so")) and int(os.environ.get("RANK", "0")) == 0:
        from jda_amd import build as lib_build      # fresh checkout: compile the library once
        lib_build.build()
    from jda_amd import api, dist as jdist, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the detect path has no CPU fallback")
    # test hooks (not used by the driver): all ranks on one GPU and a gloo group, to exercise the N>1 control
    # flow on a 1-GPU box -- RCCL itself needs one GPU per rank
    backend = os.environ.get("JDA_BENCH_BACKEND", "nccl")
    if os.environ.get("JDA_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gather_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    dims = tuple(int(x) for x in args.dims.split(","))
    T, K, L, D = dims
    W, H, B = args.width, args.height, args.batch
    call = dict(scale=1.25, min_size=40, max_size=-1, th=-0.5)

    calib = synth.make_frames(8, W, H, seed=0, first=10_000_000)
    R = max(1, args.rotate)
    # this rank's shard of the job: R distinct batches, so that a step does not re-read frames the previous step left
    # in the Infinity Cache
    frames_all = [synth.make_frames(B, W, H, seed=0, first=(rank * R + j) * B) for j in range(R)]
    frames = frames_all[0]
    d_batches = [torch.from_numpy(f).to(dev) for f in frames_all]
    d_frames = d_batches[0]
    wpf, n_levels = api.count_windows(W, H, call["scale"], call["min_size"], call["max_size"])
    windows_step = wpf * B

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def lsync():
        """Rank-local: the legs only rank 0 runs (after the job-wide timed regions) must not enter a collective."""
        torch.cuda.synchronize()

    gather_kind = ["none (one rank)"]

    gather_cache = {}

    def make_gather():
        """One gather object per process, made at first use and reused by every leg (at N ranks a libjda_dist communicator
        is an ncclCommInitRank across the node: seconds, not something to repeat per leg)."""
        key = os.environ.get("JDA_BENCH_GATHER", "c")
        if key not in gather_cache:
            gather_cache[key] = new_gather()
        g, kind = gather_cache[key]
        g.drain()
        return g, kind

    def close_gathers():
        for g, _ in gather_cache.values():
            if hasattr(g, "close"):
                g.close()
        gather_cache.clear()

    def new_gather():
        """The gather of (bbox, score, landmarks) rows on rank 0, one collective per step, pipelined one step behind the
        detection.  N > 1 over RCCL: the C entry points of libjda_dist.so (include/jda_dist.h: jdaDistGatherStart /
        jdaDistGatherCollect on the library's own communicator); if that library cannot be set up on every rank, the
        same exchange through torch.distributed (jda_amd/dist.py:PipelinedGather)."""
        width = 5 + 2 * L
        # (JDA_BENCH_C_GATHER_ON_GLOO=1 is for tests/test_dist_stub.py: two ranks on ONE GPU, torch's group over gloo and
        # the library's RCCL calls answered by the preloaded stand-in -- the bench's N>1 flow with the C gather in it)
        c_backend = backend == "nccl" or os.environ.get("JDA_BENCH_C_GATHER_ON_GLOO") == "1"
        if world > 1 and c_backend and os.environ.get("JDA_BENCH_GATHER", "c") == "c":
            ok = 1
            try:
                jdist.dist_lib()
            except Exception:
                ok = 0
            head = torch.zeros(129, dtype=torch.uint8, device=dev)
            if rank == 0 and ok:
                try:
                    head[1:] = torch.frombuffer(bytearray(jdist.unique_id()), dtype=torch.uint8).to(dev)
                    head[0] = 1
                except Exception:
                    pass
            dist.broadcast(head, 0)
            flag = torch.tensor([ok if int(head[0].item()) == 1 else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                g = None
                try:
                    g = jdist.CGather(rank, world, head[1:].cpu().numpy().tobytes(), local_rank, width, 4096)
                except Exception as e:                      # noqa: BLE001 -- (a rank that cannot join: every rank falls back)
                    sys.stderr.write("bench.py rank %d: libjda_dist communicator not set up (%r), falling back to torch.distributed\n" % (rank, e))
                made = torch.tensor([1 if g is not None else 0], device=dev)
                dist.all_reduce(made, op=dist.ReduceOp.MIN)
                if int(made.item()) == 1:
                    return g, "libjda_dist.so: jdaDistGatherStart/Collect (ncclAllGather of fixed blocks on the library's communicator)"
                if g is not None and hasattr(g, "close"):
                    g.close()
        kind = "torch.distributed all_gather (%s)" % backend if world > 1 else "none (one rank)"
        return jdist.PipelinedGather(4096, width, device=gather_dev), kind

    private = {}          # (not part of the JSON line: the last timed step's rows of the most recent run_regime)

    def run_regime(regime, steps, warmup, th, lanes=None, depth=1):
        """lanes=1 serialises the library's two sub-batch lanes and keeps the global-pixel launch on the lane's own
        stream (JDA_LANES=1, JDA_SIDE_STREAM=0): the k_scan launches then run back to back and the HIP-event span
        around them is the sum of their durations (roofline leg)."""
        keys = {"JDA_LANES": str(lanes), "JDA_SIDE_STREAM": "0"} if lanes is not None else {}
        saved = {k: os.environ.get(k) for k in keys}
        os.environ.update(keys)
        try:
            return _run_regime(regime, steps, warmup, th, depth)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    def _run_regime(regime, steps, warmup, th, depth):
        """depth 2: the K steps go through the submit/wait entry points, two batches in flight from this one thread.
        Every step is still one complete pass over one batch, and all K are finished inside the timed region."""
        mp = model_path(dims, regime, 1, calib)
        depth = max(1, min(depth, 3, steps))          # (the library has three tickets)
        cascs = [api.Cascador(mp, device=local_rank)]
        casc = cascs[0]

        # N>1: one RCCL all_gather of fixed-size blocks brings the (bbox, score, landmarks) rows to rank 0.  It is
        # pipelined one step behind: the collective of step i runs on the communicator's stream while step i+1
        # detects and is collected (counts, valid rows -> host on rank 0) by step i+1; the last one is drained
        # before the closing barrier, so all K gathers complete inside the timed region.
        gather, gather_kind[0] = make_gather()

        counter = [0]
        timed_passes = os.environ.get("JDA_BENCH_TIMED_PASSES", "0") == "1"

        def next_batch():
            counter[0] += 1
            return d_batches[counter[0] % R]

        def step(want_stats=False, c=None):
            # every rank: detect its batch, flatten the (bbox, score, landmarks) tuples with one C call
            out = (c or casc).detect_batch_device(next_batch(), call["scale"], call["min_size"], call["max_size"], th,
                                                  nms=True, stats=want_stats, keep_results="packed",
                                                  frame_offset=rank * B)
            rows, st = out if want_stats else (out, None)
            if world > 1:
                gather.start(rows)
            private["step_rows"] = rows
            return len(rows), st

        def submit():
            # (no timing events in the passes of the headline leg: the kernel spans come from the one-lane leg)
            return casc.submit_batch_device(next_batch(), call["scale"], call["min_size"], call["max_size"], th, nms=True,
                                            stats=timed_passes)

        if depth == 1:
            for _ in range(warmup):
                step()
        elif warmup > 0:                 # warm up the path that is timed (its lanes hold a whole batch each)
            q = [submit() for _ in range(min(depth - 1, warmup))]
            issued = len(q)
            for i in range(warmup):
                if issued < warmup:
                    q.append(submit()); issued += 1
                rows = casc.wait_batch(q.pop(0), keep_results="packed", frame_offset=rank * B)
                if world > 1:
                    gather.start(rows)
        gather.drain()
        barrier()
        t0 = time.perf_counter()
        stats = []
        n_det = 0
        last_rows = None
        if depth == 1:
            for _ in range(steps):
                n_det, st = step(True)
                stats.append(st)
            last_rows = (np.array(private["step_rows"], copy=True), counter[0] % R)
        else:
            # two batches in flight from this one thread: the scan of step i+1 is queued before step i is
            # collected (jdaDetectBatchSubmit / jdaDetectBatchWait), so the GPU works on it while the host parts of
            # step i run (queue-length reads, D2H, sort, NMS, result assembly, gather)
            # (depth 3: two batches are queued ahead of the one being collected)
            q = [submit() for _ in range(min(depth - 1, steps))]
            issued = len(q)
            for i in range(steps):
                if issued < steps:
                    q.append(submit()); issued += 1
                rows, st = casc.wait_batch(q.pop(0), stats=True, keep_results="packed", frame_offset=rank * B)
                if world > 1:
                    gather.start(rows)
                n_det = len(rows)
                stats.append(st)
                if i == steps - 1:
                    # what the LAST TIMED step produced, and which resident batch it ran on (submitted `steps` submits after
                    # the warm-up's): checked against the reference after the clock has stopped (parity_check below)
                    last_rows = (np.array(rows, copy=True), counter[0] % R)
        gather.drain()
        barrier()
        el = time.perf_counter() - t0
        rank_el = None
        if world > 1:
            # the job's time is the slowest rank's; every rank's own time goes into config.rank_ms_per_step_min / _max
            t = torch.tensor([el], dtype=torch.float64, device=gather_dev)
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            rank_el = [float(x.item()) for x in every]
            el = max(rank_el)
        st = stats[-1]
        scan_ms = float(np.mean([s["scan_ms"] for s in stats]))
        scan_lds_ms = float(np.mean([s["scan_lds_ms"] for s in stats]))
        scan_lds_carts = float(np.mean([s["scan_lds_cart_n"] for s in stats]))
        gpu_ms = float(np.mean([s["gpu_ms"] for s in stats]))
        host_ms = float(np.mean([s["host_ms"] for s in stats]))
        call_ms = float(np.mean([s["call_ms"] for s in stats]))
        step_bytes = algorithmic_bytes(dims, st["cart_total_n"], st["stage_done_n"][:T], st["patch_n"], n_det)
        scan_bytes = st["scan_cart_n"] * ((D - 1) * 34 + 16) + st["scan_patch_n"] * 2 * L * 4
        info = {
            "windows_per_s": windows_step * world * steps / el,
            "images_per_s": B * world * steps / el,
            "ms_per_step": el / steps * 1e3,
            "gpu_ms_per_step": gpu_ms, "scan_ms_per_step": scan_ms, "host_post_ms_per_step": host_ms,
            "scan_lds_ms_per_step": scan_lds_ms, "scan_lds_carts_per_step": scan_lds_carts,
            "scan_carts_per_step": float(np.mean([s["scan_cart_n"] for s in stats])),
            "call_ms_per_step": call_ms,
            "average_cart_n": st["average_cart_n"], "finish_fraction": st["stage_done_n"][T - 1] / max(1, st["patch_n"]),
            "detections_after_nms": n_det,
            "step_algorithmic_GBps": step_bytes / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else None,
            "scan_algorithmic_bytes": scan_bytes, "scan_launches": st["scan_launches"],
            "scan_window_fraction": st["scan_patch_n"] / max(1, st["patch_n"]),
            "dense_passes_per_step": st["dense_passes"],
            "batches_in_flight": depth,
            "rank_ms_per_step": [e / steps * 1e3 for e in rank_el] if rank_el else None,
        }
        private["last_rows"] = last_rows
        for c in cascs:
            c.close()
        return info, mp

    # ---- first contact of the N-rank gather, before anything is timed: a known ragged pattern (an empty rank, a rank with
    #      more rows than a pipelined block carries -> every rank takes the exact ncclAllGather(counts) + ncclSend/ncclRecv
    #      path, two gathers in flight) through the same entry points the timed steps use, checked row by row on rank 0.
    #      JDA_DIST_SELFTEST=0 skips it; a failure is reported in config.dist_selftest and the job goes on through
    #      whatever gather still works ----
    selftest = None
    if world > 1 and os.environ.get("JDA_DIST_SELFTEST", "1") != "0":
        g0, kind0 = make_gather()
        try:
            selftest = jdist.gather_selftest(g0, rank, world, 5 + 2 * L, 4096)
        except Exception as e:                    # noqa: BLE001
            selftest = "FAILED on rank %d: %r" % (rank, e)
        flag = torch.tensor([1 if selftest == "ok" else 0], device=gather_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            if selftest == "ok":
                selftest = "FAILED on another rank"
            sys.stderr.write("bench.py rank %d: gather self-test (%s): %s\n" % (rank, kind0, selftest))
            if "libjda_dist" in kind0:
                os.environ["JDA_BENCH_GATHER"] = "torch"      # every rank saw the same flag: all fall back together
        else:
            selftest = "ok (%s)" % kind0.split(":")[0]

    casc_info, casc_model = run_regime("cascade", args.steps, args.warmup, call["th"], depth=args.depth)
    headline_rows = private.get("last_rows")          # (rows of the last TIMED step, index of the batch it ran on)
    rank_ms = casc_info.get("rank_ms_per_step")
    # one caller, one batch at a time (what a single jdaDetectBatchDevice loop sees)
    single_info = casc_info if args.depth <= 1 else run_regime("cascade", max(1, min(20, args.steps)), 2, call["th"])[0]
    # roofline leg: the same workload with the k_scan launches of a step back to back on one stream
    roof_info, _ = run_regime("cascade", max(1, min(20, args.steps)), 2, call["th"], lanes=1)
    allpass_info = None
    if not args.no_allpass:
        # every window walks all T*K carts; final th=+inf so NMS sees nothing
        allpass_info, _ = run_regime("allpass", max(1, min(2, args.steps)), 1, float("inf"))

    # PCIe-inclusive leg (never `value`): the same steps with the frames in HOST memory, through
    # jdaDetectBatchSubmitHost / jdaDetectBatchWait (two batches in flight: the H2D copy of step i+1 runs next to the
    # kernels of step i), once from pageable numpy arrays and once from pinned ones
    def host_leg(pinned, steps):
        casc = api.Cascador(casc_model, device=local_rank)
        srcs = frames_all[:2]
        keep = []
        if pinned:
            keep = [torch.from_numpy(f).pin_memory() for f in srcs]
            srcs = [k.numpy() for k in keep]
        kw = dict(scale=call["scale"], min_size=call["min_size"], max_size=call["max_size"], th=call["th"])
        def run(k, ahead=2):
            # two batches submitted ahead of the one being collected (three tickets): a ticket lives for copy +
            # kernels + host work, about twice the copy time, so one ahead leaves the PCIe link idle in between
            q = [casc.submit_batch_host(srcs[j % 2], **kw) for j in range(min(ahead, k))]
            for i in range(k):
                if i + ahead < k:
                    q.append(casc.submit_batch_host(srcs[(i + ahead) % 2], **kw))
                casc.wait_batch(q.pop(0), keep_results="packed")
        run(3)
        lsync(); t0 = time.perf_counter()
        run(steps)
        lsync(); el = time.perf_counter() - t0
        casc.close()
        return windows_step * steps / el

    # (single-GPU legs from here on run on rank 0 only, at every N: the other ranks wait at the closing barrier, so that a
    # line printed by an N-rank job carries the same keys as the one-GPU line)
    host_info = None
    if rank == 0:
        hs = max(2, min(30, args.steps))
        host_info = {"pageable_windows_per_s": host_leg(False, hs), "pinned_windows_per_s": host_leg(True, hs), "steps": hs,
                     "entry": "jdaDetectBatchSubmitHost / jdaDetectBatchWait, three tickets (two batches submitted ahead)"}

    # ---- BASELINE.json configs[3] / the metric's "FDDB images/sec": an FDDB-shaped job (2,845 images <= 450x450 of
    #      varied aspect, the reference's one-Detect-per-image loop, src/test.cpp:100-170) as ONE ragged job
    #      (jdaDetectBatchRaggedDevice), images resident in HBM, sharded over the ranks in contiguous blocks,
    #      detections gathered on rank 0 inside the timed region.  A FIXED job: strong scaling. ----
    def fddb_set():
        """This rank's contiguous block of the FDDB-shaped image set (2,845 images <= 450x450), built once."""
        if "fddb_set" in private:
            return private["fddb_set"]
        n_img = 2845
        rng = np.random.default_rng(0)
        sizes = []
        for _ in range(n_img):
            long_side = int(rng.integers(300, 451)); short = int(rng.integers(225, long_side + 1))
            sizes.append((long_side, short) if rng.random() < 0.5 else (short, long_side))
        # contiguous blocks of about equal WORK: an image weighs its candidate windows (c/jda.c:320-339 via jdaCountWindows);
        # every rank computes the same boundaries (jda_amd/dist.py:shard_range_weighted)
        weights = np.array([api.count_windows(w_, h_, call["scale"], call["min_size"], call["max_size"])[0] for (w_, h_) in sizes], np.float64)
        private["fddb_weights"] = weights
        lo, hi = jdist.shard_range_weighted(weights, rank, world)
        base = synth.make_frames(64, 450, 450, seed=7)          # images = crops of 64 synthetic 450x450 frames
        imgs = [np.ascontiguousarray(base[i % 64][:sizes[i][1], :sizes[i][0]]) for i in range(lo, hi)]
        offs, tot = [], 0
        for im in imgs:
            offs.append(tot); tot += im.size
        buf = np.concatenate([im.reshape(-1) for im in imgs]) if imgs else np.zeros(0, np.uint8)
        # (the job description as arrays: the binding hands int32 / uint64 arrays to the C entry as they are -- lists of
        # thousands of Python ints cost 15 us each to convert, per call)
        ws, hs = np.array([sizes[i][0] for i in range(lo, hi)], np.int32), np.array([sizes[i][1] for i in range(lo, hi)], np.int32)
        offs = np.array(offs, np.uint64)
        d_buf = torch.from_numpy(buf).to(dev)
        private["fddb_set"] = (n_img, sizes, lo, hi, imgs, offs, tot, buf, ws, hs, d_buf)
        return private["fddb_set"]

    def fddb_leg(reps):
        n_img, sizes, lo, hi, imgs, offs, tot, buf, ws, hs, d_buf = fddb_set()
        casc = api.Cascador(casc_model, device=local_rank)
        gather, _ = make_gather()
        kw = dict(scale=call["scale"], min_size=call["min_size"], max_size=call["max_size"], th=call["th"])

        def job(src):
            rows, st = casc.detect_ragged_packed(src, offs, ws, hs, stats=True, keep_results="packed", frame_offset=lo, **kw)
            if world > 1:
                gather.start(rows)
            return rows, st
        for _ in range(2):
            job(d_buf)
        gather.drain()
        barrier(); t0 = time.perf_counter()
        for _ in range(reps):
            rows, st = job(d_buf)
        gather.drain()
        barrier(); el = time.perf_counter() - t0
        windows = st["patch_n"]
        if world > 1:
            t = torch.tensor([el, float(windows)], dtype=torch.float64, device=gather_dev)
            tm = t.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX); el = float(tm[0].item())
            dist.all_reduce(t, op=dist.ReduceOp.SUM); windows = int(t[1].item())
        private["fddb_rows"] = (np.array(rows, copy=True), lo, imgs)      # the last TIMED job's rows (parity_check below)
        info = {"images": n_img, "images_per_s": n_img * reps / el, "windows_per_s": windows * reps / el,
                "ms_per_job": el / reps * 1e3, "jobs_timed": reps, "windows_per_job": windows, "scaling": "strong",
                "entry": "jdaDetectBatchRaggedDevice (images resident in HBM), one host thread per GPU",
                "sharding": "contiguous blocks of images of about equal candidate-window counts over %d rank(s)" % world,
                "data": "synthetic, FDDB-like sizes (long side 300-450, short side >= 225)"}
        if world == 1:
            job(buf)
            t0 = time.perf_counter()
            for _ in range(reps):
                job(buf)
            info["host_images_per_s"] = n_img * reps / (time.perf_counter() - t0)
            info["host_entry"] = "jdaDetectBatchRagged, the images in one pageable host buffer (PCIe-inclusive)"
            # What ONE GPU can say about the N-GPU job (the driver measures N > 1 itself when it has the node): every
            # rank's shard -- the contiguous block jda_amd/dist.py:shard_range gives rank r of N -- run alone on this
            # GPU, images resident, best of `reps`; the job on N GPUs takes as long as its slowest shard plus the gather
            # of the rows (KBs over xGMI, pipelined one step behind: not on the critical path of a stream of jobs).
            # predicted speed-up = t(all 2,845 images) / max_r t(shard r).
            t_full = el / reps
            pred = {}
            for nn in (2, 4, 8):
                worst, per = 0.0, []
                for r in range(nn):
                    a, b = jdist.shard_range_weighted(private["fddb_weights"], r, nn)
                    base_off = int(offs[a])
                    so = offs[a:b] - np.uint64(base_off)
                    end = int(offs[b]) if b < n_img else tot
                    d_sub = d_buf[base_off:end]
                    ws_s, hs_s = np.ascontiguousarray(ws[a:b]), np.ascontiguousarray(hs[a:b])

                    def shard_job():
                        return casc.detect_ragged_packed(d_sub, so, ws_s, hs_s, stats=True, keep_results="packed", frame_offset=a, **kw)
                    shard_job(); shard_job()
                    nrep = max(11, reps)
                    ts = []
                    for _ in range(nrep):
                        t1 = time.perf_counter()
                        shard_job()
                        ts.append(time.perf_counter() - t1)
                    med = float(np.median(ts))          # (the median of the runs: one descheduled host thread does not define a shard)
                    per.append(med); worst = max(worst, med)
                pred[str(nn)] = {"max_shard_ms": worst * 1e3, "min_shard_ms": min(per) * 1e3, "speedup": t_full / worst,
                                 "shard_ms": [round(t * 1e3, 4) for t in per]}
            info["predicted_strong_scaling"] = pred
            info["predicted_strong_scaling_note"] = ("each rank's shard of the job timed alone on this GPU (median of %d runs after two warm-ups); speedup = "
                                                     "mean ms_per_job / slowest shard's median; not a multi-GPU measurement" % max(11, reps))
        casc.close()
        return info

    fddb_info = None
    try:
        fddb_info = fddb_leg(max(1, min(5, args.steps)))
    except Exception as e:
        if world > 1:
            raise
        fddb_info = {"error": repr(e)}

    # ---- dialect CPP, the dialect the reference's own `jda fddb` runs: joincascador.Detect with fddb.method = 1
    #      (src/test.cpp:142; model/config.json:41-45: minimum_size 20, step 5, scale 1.2, overlap 0.3, nms on).  Two legs,
    #      both with the data resident in HBM: the headline batch (256 x 640x480: 140,215 windows per frame) through
    #      jdaDetectBatchCppDevice, and the FDDB-shaped job as ONE jdaDetectBatchCppRaggedDevice job per rank (strong
    #      scaling over the ranks, float64 rows gathered on rank 0 inside the timed region).  PARITY UNPINNED:
    #      the fp64 src/jda path cannot be compiled here; checked against the oracle's restatement after the clock stops. ----
    CPP = dict(minimum_size=20, step=5, factor=1.2, overlap=0.3, nms=True)

    def cpp_leg(reps):
        casc = api.Cascador(casc_model, device=local_rank)
        wpf_cpp = synth.count_windows_cpp(W, H, CPP["minimum_size"], CPP["step"], CPP["factor"])
        for j in range(2):
            casc.detect_batch_cpp_device(d_batches[j % R], keep_results=False, **CPP)
        barrier(); t0 = time.perf_counter()
        for i in range(reps):
            rows, st = casc.detect_batch_cpp_device(d_batches[i % R], stats=True, keep_results="packed", frame_offset=rank * B, **CPP)
        barrier(); el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=gather_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX); el = float(t[0].item())
        private["cpp_rows"] = (np.array(rows, copy=True), (reps - 1) % R)
        casc.close()
        return {"windows_per_s": wpf_cpp * B * world * reps / el, "images_per_s": B * world * reps / el, "ms_per_step": el / reps * 1e3,
                "windows_per_frame": wpf_cpp, "gpu_ms_per_step": st["gpu_ms"], "host_post_ms_per_step": st["host_ms"],
                "average_cart_n": st["average_cart_n"], "face_patch_n": st["face_patch_n"], "detections_after_nms": len(rows),
                "steps": reps, "scaling": "weak", "dtype": "f64",
                "entry": "jdaDetectBatchCppDevice (frames resident in HBM), synchronous, two sub-batch lanes",
                "call": "Detect, method 1: minimum_size 20, step 5, factor 1.2, overlap 0.3, nms (model/config.json:41-45)",
                "parity": "unpinned (oracle restatement only)"}

    def fddb_cpp_leg(reps):
        from jda_amd import fddb as jfddb
        n_img, sizes, lo, hi, imgs, offs, tot, buf, ws, hs, d_buf = fddb_set()
        casc = api.Cascador(casc_model, device=local_rank)

        def job(src):
            rows, st = casc.detect_ragged_cpp_packed(src, offs, ws, hs, stats=True, keep_results="packed", frame_offset=lo, **CPP)
            if world > 1:
                jfddb._gather64(rows, gather_dev)       # float64 rows -> rank 0 (torch.distributed: counts, then padded blocks)
            return rows, st
        for _ in range(2):
            job(d_buf)
        barrier(); t0 = time.perf_counter()
        for _ in range(reps):
            rows, st = job(d_buf)
        barrier(); el = time.perf_counter() - t0
        windows = st["patch_n"]
        if world > 1:
            t = torch.tensor([el, float(windows)], dtype=torch.float64, device=gather_dev)
            tm = t.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX); el = float(tm[0].item())
            dist.all_reduce(t, op=dist.ReduceOp.SUM); windows = int(t[1].item())
        private["fddb_cpp_rows"] = (np.array(rows, copy=True), lo, imgs)
        info = {"images": n_img, "images_per_s": n_img * reps / el, "windows_per_s": windows * reps / el,
                "ms_per_job": el / reps * 1e3, "jobs_timed": reps, "windows_per_job": windows, "scaling": "strong", "dtype": "f64",
                "entry": "jdaDetectBatchCppRaggedDevice (images resident in HBM), one host thread per GPU",
                "sharding": "contiguous blocks of images of about equal dialect-C candidate-window counts over %d rank(s)" % world,
                "gather": "torch.distributed (float64 rows)" if world > 1 else "none (one rank)",
                "face_patch_n": st["face_patch_n"], "average_cart_n": st["average_cart_n"],
                "parity": "unpinned (oracle restatement only)"}
        if world == 1:
            job(buf)
            t0 = time.perf_counter()
            for _ in range(reps):
                job(buf)
            info["host_images_per_s"] = n_img * reps / (time.perf_counter() - t0)
            # the reference's structure, one Detect per image (src/test.cpp:100-170), through the per-image entry
            k = min(100, len(imgs))
            t0 = time.perf_counter()
            for i in range(k):
                casc.detect_batch_cpp(imgs[i][None], **CPP)
            info["per_image_loop_images_per_s"] = k / (time.perf_counter() - t0)
        casc.close()
        return info

    cpp_info = fddb_cpp_info = None
    try:
        cpp_info = cpp_leg(max(1, min(5, args.steps)))
        fddb_cpp_info = fddb_cpp_leg(max(1, min(5, args.steps)))
    except Exception as e:
        if world > 1:
            raise
        cpp_info = cpp_info or {"error": repr(e)}
        fddb_cpp_info = fddb_cpp_info or {"error": repr(e)}

    # ---- BASELINE.json configs[2] at its stated size, live: 256 x 1920x1080 frames resident in HBM (531 MB), scale 1.5
    #      (8 window sizes), shipped model dimensions, cascade regime.  The frames are 32 synthesised ones (synth.make_frames,
    #      1.3 s of host time instead of 10 s for 256) and seven cyclic shifts of each, made on the device: 256 distinct
    #      frames of the same statistics at 256 distinct places in HBM ----
    config2_live = None
    if rank == 0 and not args.no_config2:
        try:
            n_base, n_var = 32, 8
            f2 = synth.make_frames(n_base, 1920, 1080, seed=0)
            mp2 = os.path.join(synth.cache_dir(), "config2_5_540_27_4.model")
            if not os.path.exists(mp2):
                m2 = synth.make_model(5, 540, 27, 4, seed=1)
                synth.calibrate_thresholds(m2, f2[:4], scale=1.5)
                m2.save(mp2 + ".tmp", 8); os.replace(mp2 + ".tmp", mp2)
            c2 = api.Cascador(mp2, device=local_rank)
            b2 = torch.from_numpy(f2).to(dev)
            d2 = torch.cat([b2] + [torch.roll(b2, shifts=(131 * j, 257 * j), dims=(1, 2)) for j in range(1, n_var)]).contiguous()
            del b2
            for _ in range(2):
                c2.detect_batch_device(d2, 1.5, keep_results=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                rows2, st2 = c2.detect_batch_device(d2, 1.5, keep_results="packed", stats=True)
            torch.cuda.synchronize(); el2 = (time.perf_counter() - t0) / 3
            private["config2_rows"] = (np.array(rows2, copy=True), f2[0].copy(), mp2)      # frame 0 of the last timed call (parity_check)
            config2_live = {"ms_per_call": el2 * 1e3, "windows_per_s": st2["patch_n"] / el2, "gpu_ms": st2["gpu_ms"],
                            "scan_ms": st2["scan_ms"], "average_cart_n": st2["average_cart_n"],
                            "windows_per_call": st2["patch_n"], "frames": int(d2.shape[0]),
                            "entry": "jdaDetectBatchDevice, synchronous (two sub-batch lanes)",
                            "data": "%d synthesised 1920x1080 frames + %d cyclic shifts of each (made on the device)" % (n_base, n_var - 1)}
            # the same batches through submit / wait, two queued ahead of the one being collected; warmed up in this mode:
            # a ticket's lane holds a whole batch
            def pipe2(k):
                q = [c2.submit_batch_device(d2, 1.5) for _ in range(min(2, k))]
                issued = len(q)
                for _ in range(k):
                    if issued < k:
                        q.append(c2.submit_batch_device(d2, 1.5)); issued += 1
                    c2.wait_batch(q.pop(0), keep_results=False)
            pipe2(4)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pipe2(8)
            torch.cuda.synchronize(); elp = (time.perf_counter() - t0) / 8
            config2_live["submit_wait_ms_per_call"] = elp * 1e3
            config2_live["submit_wait_windows_per_s"] = st2["patch_n"] / elp
            c2.close(); del d2, f2
            torch.cuda.empty_cache()
        except Exception as e:
            config2_live = {"error": repr(e)}

    # ---- the regime of this path in which HBM / Infinity Cache traffic is the bound: BASELINE.json configs[4] (T=7, K=2000,
    #      68 landmarks, depth 6: W = 243.7 MB, 34.8 MB per stage) with every window of a 1080p frame walking all 14,000
    #      carts -- each window gathers K 544-byte weight rows per stage, far more than L2 holds ----
    def x_leg(T_x, fw, fh, traffic_json):
        """All-pass gather regime: T_x stages of K=2000, 68 landmarks, depth 6; every window of one fw x fh frame walks every
        cart through k_finish and gathers K 544-byte weight rows per stage."""
        xd = (T_x, 2000, 68, 6)
        xp = os.path.join(synth.cache_dir(), "x_allpass.model" if T_x == 7 else "x_allpass_T%d.model" % T_x)
        if not os.path.exists(xp):
            synth.make_model(*xd, seed=2).save(xp + ".tmp", 4)
            os.replace(xp + ".tmp", xp)
        xc = api.Cascador(xp, "float", device=local_rank)
        xf = torch.from_numpy(synth.make_frames(1, fw, fh, seed=4)).to(dev)
        for _ in range(2):
            xc.detect_batch_device(xf, th=float("inf"), keep_results=False)
        lsync(); t0 = time.perf_counter()
        _, xs = xc.detect_batch_device(xf, th=float("inf"), keep_results=False, stats=True)
        lsync(); xel = time.perf_counter() - t0
        xc.close()
        xalg = algorithmic_bytes(xd, xs["cart_total_n"], xs["stage_done_n"][:T_x], xs["patch_n"], 0)
        xrows = sum(xs["stage_done_n"][:T_x]) * xd[1] * 2 * xd[2] * 4        # the weight rows alone: K rows of 2L floats per window and stage
        w_mb = T_x * 2000 * 32 * 136 * 4 / 1e6
        info = {"workload": "T=%d K=2000 L=68 D=6 (W = %.1f MB, %s the 256 MB Infinity Cache), one %dx%d frame, canonical call, every cart "
                            "threshold -inf (all-pass): %d windows x %d carts" % (T_x, w_mb, "inside" if w_mb < 256 else "BEYOND", fw, fh, xs["patch_n"], T_x * 2000),
                "w_bytes": w_mb * 1e6, "ms_per_step": xel * 1e3, "windows_per_s": xs["patch_n"] / xel, "carts_per_s": xs["cart_total_n"] / xel,
                "bound": "hbm", "achieved": xalg / xel / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": xalg / xel / 1e9 / HBM_PEAK_GBPS,
                "weight_rows_GBps": xrows / xel / 1e9, "frac_weight_rows": xrows / xel / 1e9 / HBM_PEAK_GBPS,
                "algorithmic_bytes_per_window": xalg / max(1, xs["patch_n"]),
                "traffic": None, "traffic_source": None,
                "note": "achieved = SURVEY 8(d) algorithmic bytes (10.2 MB per window: 186 B per cart + K weight rows per stage) / wall time of "
                        "one jdaDetectBatchDevice call; frac_weight_rows prices the weight rows alone (the 186 B per cart of node / pixel "
                        "bytes are served by L1 / L2 / LDS); a stage's rows (34.8 MB) exceed L2 (4 MB per XCD)"}
        xt = os.path.join(ROOT, "profiles", traffic_json)
        if os.path.exists(xt):
            tj = json.load(open(xt))
            info["traffic"] = tj["traffic_line_bytes"]
            info["traffic_over_algorithmic"] = tj["traffic_line_bytes"] / tj["algorithmic_bytes"]
            info["traffic_over_weight_rows"] = tj["traffic_line_bytes"] / tj["weight_row_bytes"]
            info["traffic_GBps_in_profiled_run"] = tj["traffic_line_bytes"] / tj["duration_s"] / 1e9
            info["traffic_useful_equivalent"] = tj["traffic_useful_equivalent_bytes"]
            info["traffic_source"] = ("profiles/%s: %s -- builder-run rocprofv3 --pmc passes, NOT measured in this run: "
                                      "fabric read requests of the k_finish dispatch x 128 B (= FETCH_SIZE x 2), FETCH_SIZE calibrated on a gather of "
                                      "544-byte rows of known size; Infinity-Cache hits are included, no counter separates them from HBM reads"
                                      % (traffic_json, tj.get("source", "?")))
            # the counters were taken on a particular build: is it the device code this run uses?
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import pmc_traffic
                info["traffic_from_this_device_code"] = tj.get("kernel_sources_sha256") == pmc_traffic.kernel_sources_sha256()
            except Exception:
                info["traffic_from_this_device_code"] = None
        return info

    x_info = x_big_info = None
    if rank == 0 and not args.no_allpass and not args.no_x:
        try:
            x_info = x_leg(7, 1920, 1080, "x_allpass_traffic.json")             # BASELINE configs[4]
        except Exception as e:
            x_info = {"error": repr(e)}
        try:
            # the same model with twice the stages: W = 487 MB cannot sit in the Infinity Cache (a quarter-size frame keeps the
            # leg short) -- is the configs[4] figure an HBM figure or a cache figure?
            x_big_info = x_leg(14, 960, 540, "x_allpass_T14_traffic.json")
        except Exception as e:
            x_big_info = {"error": repr(e)}

    cpu = None
    if rank == 0 and not args.no_cpu:
        try:
            cpu = cpu_baseline(casc_model, dims, frames)
        except Exception as e:  # the checker is optional for the measurement itself
            cpu = {"value": None, "unit": "windows/s", "cores": 0, "kind": "unavailable", "sample": repr(e)}

    # ---- did the timed regions compute the right thing?  After every clock has stopped, rank 0 compares what the LAST
    #      TIMED call of each leg returned -- three frames of the headline batch, three images of the FDDB job, one frame of
    #      configs[2] -- bit for bit with the compiled reference c/jda.c (kind "reference") or, where no build of it exists
    #      for the dimensions, the oracle's restatement ("port"); the dialect-CPP legs with the oracle (unpinned).  The
    #      checker is test infrastructure: nothing inside a timed region touches it. ----
    parity = None
    if rank == 0 and not args.no_cpu:
        parity = {"kind": None, "ok": True, "legs": {}}
        try:
            kind, detect = make_checker(casc_model, dims)
            parity["kind"] = kind
            cargs = (call["scale"], call["min_size"], call["max_size"], call["th"])
            if headline_rows is not None:
                rows, j = headline_rows
                fs = sorted({0, B // 2, B - 1})
                parity["legs"]["headline"] = all(rows_equal(rows, rank * B + f, detect(frames_all[j][f], *cargs)) for f in fs)
                parity["headline_frames"] = [int(f) for f in fs]
                parity["headline_rows_in_last_timed_step"] = int(len(rows))
            if private.get("fddb_rows") is not None:
                rows, lo, imgs = private["fddb_rows"]
                idx = sorted({0, len(imgs) // 2, len(imgs) - 1}) if imgs else []
                parity["legs"]["fddb"] = all(rows_equal(rows, lo + i, detect(imgs[i], *cargs)) for i in idx)
            if private.get("config2_rows") is not None:
                rows2, frame0, mp2 = private["config2_rows"]
                kind2, detect2 = make_checker(mp2, dims)
                parity["legs"]["config2"] = rows_equal(rows2, 0, detect2(frame0, 1.5, call["min_size"], call["max_size"], call["th"]))
                parity["config2_kind"] = kind2
            from oracle import pyoracle
            if private.get("cpp_rows") is not None or private.get("fddb_cpp_rows") is not None:
                orc = pyoracle.Oracle(casc_model)
                cppa = (CPP["minimum_size"], CPP["step"], CPP["factor"], CPP["overlap"], CPP["nms"])
                if private.get("cpp_rows") is not None:
                    rows, j = private["cpp_rows"]
                    parity["legs"]["cpp"] = all(rows_equal_d(rows, rank * B + f, orc.detect_cpp(frames_all[j][f], *cppa)) for f in (0, B - 1))
                if private.get("fddb_cpp_rows") is not None:
                    rows, lo, imgs = private["fddb_cpp_rows"]
                    idx = sorted({0, len(imgs) // 2, len(imgs) - 1}) if imgs else []
                    parity["legs"]["fddb_cpp"] = all(rows_equal_d(rows, lo + i, orc.detect_cpp(imgs[i], *cppa)) for i in idx)
            parity["ok"] = all(parity["legs"].values()) and len(parity["legs"]) > 0
        except Exception as e:                      # noqa: BLE001 -- (the checker is optional for the measurement itself)
            parity = {"kind": "unavailable", "ok": None, "error": repr(e), "legs": parity.get("legs", {})}
        if parity.get("ok") is False:
            sys.stderr.write("bench.py: PARITY CHECK FAILED: %r\n" % (parity,))

    if world > 1:
        dist.barrier()          # (the other ranks have waited here while rank 0 ran its single-GPU legs and the checks)

    if rank == 0:
        # ---- roofline of the dominant kernel: the LDS-tiled k_scan launches of one step ----
        # Bound: the LDS pipe.  Every tree node costs three lane-reads from LDS (one 8-byte resolved node record, two
        # pixel bytes), every cart two more (leaf score, threshold): (D-1)*3 + 2 lane-reads per window-cart; a lane-read
        # moves one 4-byte bank word at most.  achieved = lane-reads of the carts the launches evaluated (device counter)
        # x 4 B / the HIP-event span of those launches (on their own stream, back to back).
        lane_reads_per_cart = (D - 1) * 3 + 2
        lds_s = roof_info["scan_lds_ms_per_step"] * 1e-3
        lane_reads = roof_info["scan_lds_carts_per_step"] * lane_reads_per_cart
        achieved = lane_reads * 4 / lds_s / 1e9 if lds_s > 0 else 0.0
        traffic, traffic_src, hbm_frac, traffic_fresh = None, None, None, None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                traffic = tj.get("k_scan_lds_bytes_per_step", tj.get("k_scan_bytes_per_step"))
                traffic_src = ("profiles/hbm_traffic.json: %s -- builder-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                               "this command, NOT measured in this run" % tj.get("source", "r01"))
                if traffic and lds_s > 0:
                    hbm_frac = traffic / lds_s / 1e9 / HBM_PEAK_GBPS
                # the counters were taken on a particular build: say whether it is the device code this run uses
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import pmc_traffic
                traffic_fresh = tj.get("kernel_sources_sha256") == pmc_traffic.kernel_sources_sha256()
            except Exception:
                traffic = None
        scan_bytes_alg = roof_info["scan_lds_carts_per_step"] * ((D - 1) * 34 + 16)
        # ---- the dense path (k_stage: every window walks every cart and takes every cart's weight row, all from LDS tables
        #      staged once per 256-window tile): LDS crossbar bytes per window-cart = the walk's lane-reads + the 2L floats of
        #      the weight row.  The pipe-busy fractions are counter readings of the same kernel (profiles/r04_allpass_k_stage.txt,
        #      builder-run): neither pipe is saturated -- the kernel waits on LDS latency at the occupancy its tables allow ----
        allpass_roof = None
        if allpass_info and allpass_info.get("gpu_ms_per_step"):
            carts_s = allpass_info["average_cart_n"] * windows_step / (allpass_info["gpu_ms_per_step"] * 1e-3)
            reads = lane_reads_per_cart + 2 * L
            ach = carts_s * reads * 4 / 1e9
            allpass_roof = {"bound": "lds", "achieved": ach, "peak": LDS_PEAK_GBPS, "unit": "GB/s", "frac": ach / LDS_PEAK_GBPS,
                            "kernel": "k_stage (dense mode), shipped dimensions, all-pass regime",
                            "carts_per_s": carts_s, "lane_reads_per_cart": reads,
                            "what": "LDS crossbar bytes: (%d walk + %d weight-row) lane-reads of 4 B per window-cart x carts / device span of the step" % (lane_reads_per_cart, 2 * L),
                            "counters_in": "profiles/r04_allpass_k_stage.txt (builder-run rocprofv3 --pmc SQ passes of k_stage: LDS pipe and VALU busy fractions; not echoed here)"}
        # ---- flat scalars under `config` and `roofline`: the two objects a reader of the driver's record keeps ----
        def g(d, *ks):
            for k in ks:
                if not isinstance(d, dict) or d.get(k) is None:
                    return None
                d = d[k]
            return d
        cfg_extra = {
            "images_per_s": casc_info["images_per_s"],
            # BASELINE metric, second half: the FDDB-shaped job (configs[3]'s 2,845 images as one ragged job)
            "fddb_images_per_s": g(fddb_info, "images_per_s"), "fddb_ms_per_job": g(fddb_info, "ms_per_job"),
            "fddb_windows_per_s": g(fddb_info, "windows_per_s"), "fddb_images": g(fddb_info, "images"),
            "fddb_jobs_timed": g(fddb_info, "jobs_timed"), "fddb_host_images_per_s": g(fddb_info, "host_images_per_s"),
            "fddb_pred_speedup_2": g(fddb_info, "predicted_strong_scaling", "2", "speedup"),
            "fddb_pred_speedup_4": g(fddb_info, "predicted_strong_scaling", "4", "speedup"),
            "fddb_pred_speedup_8": g(fddb_info, "predicted_strong_scaling", "8", "speedup"),
            # configs[2]: 256 x 1080p, scale 1.5, measured in this run
            "config2_windows_per_s": g(config2_live, "windows_per_s"), "config2_ms_per_call": g(config2_live, "ms_per_call"),
            "config2_gpu_ms_per_call": g(config2_live, "gpu_ms"), "config2_windows_per_call": g(config2_live, "windows_per_call"),
            "config2_frames": g(config2_live, "frames"),
            "config2_submit_wait_windows_per_s": g(config2_live, "submit_wait_windows_per_s"),
            "config2_submit_wait_ms_per_call": g(config2_live, "submit_wait_ms_per_call"),
            # all-pass regime of the headline batch (dense kernel) and configs[4] (one 1080p frame, all-pass)
            "allpass_windows_per_s": g(allpass_info, "windows_per_s"), "allpass_ms_per_step": g(allpass_info, "ms_per_step"),
            "config4_allpass_windows_per_s": g(x_info, "windows_per_s"), "config4_allpass_ms_per_frame": g(x_info, "ms_per_step"),
            "one_lane_ms_per_step": roof_info["ms_per_step"], "average_cart_n": casc_info["average_cart_n"],
            "detections_after_nms": casc_info["detections_after_nms"],
            # dialect CPP -- the reference's own fddb() dialect (Detect, method 1), resident data, measured in this run
            "cpp_windows_per_s": g(cpp_info, "windows_per_s"), "cpp_ms_per_step": g(cpp_info, "ms_per_step"),
            "cpp_windows_per_frame": g(cpp_info, "windows_per_frame"),
            "fddb_cpp_images_per_s": g(fddb_cpp_info, "images_per_s"), "fddb_cpp_ms_per_job": g(fddb_cpp_info, "ms_per_job"),
            "fddb_cpp_windows_per_s": g(fddb_cpp_info, "windows_per_s"),
            "fddb_cpp_per_image_loop_images_per_s": g(fddb_cpp_info, "per_image_loop_images_per_s"),
            "cpp_parity": "unpinned (fp64 src/jda path not compilable here; checked against the oracle's restatement)",
            # what the last timed call of each leg returned, compared bit for bit after the clocks stopped: "reference" =
            # the compiled c/jda.c, "port" = the oracle's restatement, False = a mismatch, None = no checker available
            "parity_checked": (parity["kind"] if parity and parity.get("ok") else (False if parity and parity.get("ok") is False else None)),
            "parity_legs": g(parity, "legs"),
        }
        if rank_ms is not None:
            cfg_extra.update({"rank_ms_per_step_min": min(rank_ms), "rank_ms_per_step_max": max(rank_ms)})
        if selftest is not None:
            cfg_extra["dist_selftest"] = selftest
        roof_extra = {
            # the regime in which memory traffic IS the bound (configs[4] all-pass), measured in this run
            "hbm_regime_bound": "hbm", "hbm_regime_frac": g(x_info, "frac"), "hbm_regime_achieved_GBps": g(x_info, "achieved"),
            "hbm_regime_peak_GBps": HBM_PEAK_GBPS, "hbm_regime_ms_per_step": g(x_info, "ms_per_step"),
            "hbm_regime_traffic_over_algorithmic": g(x_info, "traffic_over_algorithmic"),
            "hbm_regime_frac_weight_rows": g(x_info, "frac_weight_rows"),
            "hbm_regime_traffic_from_this_device_code": g(x_info, "traffic_from_this_device_code"),
            # the same gather with W = 487 MB, beyond the 256 MB Infinity Cache (T=14, one 960x540 frame), measured in this run
            "hbm_regime_beyond_mall_frac": g(x_big_info, "frac"), "hbm_regime_beyond_mall_frac_weight_rows": g(x_big_info, "frac_weight_rows"),
            "hbm_regime_beyond_mall_ms_per_step": g(x_big_info, "ms_per_step"), "hbm_regime_beyond_mall_w_bytes": g(x_big_info, "w_bytes"),
            "hbm_regime_beyond_mall_traffic_over_algorithmic": g(x_big_info, "traffic_over_algorithmic"),
            "hbm_regime_workload": "configs[4]: T=7 K=2000 L=68 D=6, one 1080p frame, all-pass: 303,222 windows x 14,000 carts",
            "allpass_lds_frac": g(allpass_roof, "frac"), "allpass_lds_achieved_GBps": g(allpass_roof, "achieved"),
            "hbm_side_measured_frac_of_hbm_peak": hbm_frac,
            "hbm_side_algorithmic_GBps": scan_bytes_alg / lds_s / 1e9 if lds_s > 0 else None,
        }
        line = {
            "metric": "candidate windows/sec, 640x480 batch (jdaDetect hot path); FDDB images/sec in config.fddb_images_per_s",
            "value": casc_info["windows_per_s"], "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": casc_info["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": dict({"workload": "BASELINE.json configs[1]: batch=%d %dx%d frames per GPU, synthetic %dx%d-cart "
                                        "%d-landmark depth-%d model, cascade regime, jdaDetect(1.25,0.1,40,-1,-0.5)"
                                        % (B, W, H, T, K, L, D),
                            "batch_per_gpu": B, "width": W, "height": H, "windows_per_frame": wpf, "levels": n_levels,
                            "model_T": T, "model_K": K, "model_landmarks": L, "model_depth": D,
                            "regime": "cascade", "sharding": "frames, %d rank(s)" % world,
                            "gather": gather_kind[0],
                            "distinct_resident_batches": R, "resident_frame_bytes": R * B * W * H,
                            "batches_in_flight_per_gpu": casc_info["batches_in_flight"],
                            "single_caller_ms_per_step": single_info["ms_per_step"],
                            "single_caller_windows_per_s": single_info["windows_per_s"],
                            "host_frames_windows_per_s": host_info["pageable_windows_per_s"] if host_info else None,
                            "host_frames_pinned_windows_per_s": host_info["pinned_windows_per_s"] if host_info else None,
                            "host_frames_pinned_over_value": (host_info["pinned_windows_per_s"] / casc_info["windows_per_s"])
                                                             if host_info else None}, **cfg_extra),
            "roofline": dict({"bound": "lds", "achieved": achieved, "peak": LDS_PEAK_GBPS,
                              "unit": "GB/s", "frac": achieved / LDS_PEAK_GBPS,
                              "traffic": traffic, "traffic_source": traffic_src,
                              "traffic_from_this_device_code": traffic_fresh,
                              "kernel": "k_scan_p / k_scan, LDS-tiled launches of one step (one per tiled pyramid level)",
                              "what": "LDS crossbar bytes: %d lane-reads x 4 B per window-cart x carts (device counter) / HIP-event span"
                                      % lane_reads_per_cart,
                              "lane_reads_per_cart": lane_reads_per_cart,
                              "carts_per_step": roof_info["scan_lds_carts_per_step"],
                              "kernel_ms_per_step": roof_info["scan_lds_ms_per_step"],
                              "frac_of_random_gather_rate": achieved / LDS_RANDOM_GATHER_GBPS,
                              "random_gather_rate_GBps": LDS_RANDOM_GATHER_GBPS,
                              "all_scan_launches_ms_per_step": roof_info["scan_ms_per_step"],
                              "measured_with": "JDA_LANES=1 JDA_SIDE_STREAM=0: launches back to back on one stream, HIP events on that stream"},
                             **roof_extra),
            "cpu_baseline": cpu,
            # ---- detail (a reader of the full line; the driver's record keeps config / roofline / cpu_baseline) ----
            "fddb": fddb_info,
            "cpp": cpp_info,
            "fddb_cpp": fddb_cpp_info,
            "parity": parity,
            "config2": config2_live,
            "roofline_hbm_regime": x_info,
            "roofline_hbm_regime_beyond_mall": x_big_info,
            "roofline_allpass": allpass_roof,
            "regimes": {"cascade": casc_info, "cascade_single_caller": single_info, "cascade_one_lane": roof_info,
                        "allpass": allpass_info, "host_frames": host_info},
        }
        print(json.dumps(line))
    close_gathers()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
