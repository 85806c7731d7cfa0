"""jdaDetect is re-entrant on ONE cascador, like the reference's (no globals, no locks: c/jda.c:443-480, SURVEY 8b
"Threading"): concurrent callers each take a lane (stream + workspace) from the cascador's pool; the model and the
scan plans are shared."""
import os
import threading
import time

import numpy as np
import pytest

from conftest import S_DIMS, same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda", 0)


def _eq(a, b, what=""):
    for k in ("bboxes", "scores", "shapes"):
        assert same(a[k], b[k]), (what, k)


def test_concurrent_callers_share_one_cascador(built, gpu, tmp_path):
    """8 threads call jdaDetect on one cascador (single 640x480 frames, shipped dimensions, cascade regime): every
    result equals the single-threaded one, and the call rate is at least 2.2x that of one thread (2.6-3.3x measured:
    the eight lanes' streams share HIP's four hardware queues, hence the spread)."""
    from jda_amd import api, synth
    frames = synth.make_frames(16, 640, 480, seed=3)
    m = synth.make_model(*S_DIMS, seed=1)
    synth.calibrate_thresholds(m, synth.make_frames(8, 640, 480, seed=0, first=10_000_000))
    p = str(tmp_path / "s.model"); m.save(p, 8)
    c = api.Cascador(p)
    want = [c.detect(f) for f in frames]
    reps = 40
    t0 = time.perf_counter()
    for r in range(reps):
        c.detect(frames[r % 16])
    one = reps / (time.perf_counter() - t0)
    n_thr = 8
    errors = []
    got = [[None] * reps for _ in range(n_thr)]

    def work(t):
        try:
            for r in range(reps):
                got[t][r] = c.detect(frames[(t + r) % 16])
        except Exception as e:           # noqa: BLE001
            errors.append(repr(e))
    for _ in range(2):                   # (the first round creates the lanes)
        ths = [threading.Thread(target=work, args=(t,)) for t in range(n_thr)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        many = n_thr * reps / (time.perf_counter() - t0)
    assert not errors, errors
    for t in range(n_thr):
        for r in range(reps):
            _eq(got[t][r], want[(t + r) % 16], (t, r))
    print("jdaDetect calls/s: 1 thread %.0f, %d threads on one cascador %.0f (%.1fx)" % (one, n_thr, many, many / one))
    # (2.6-3.3x measured over six boxes; 8 lanes share four hardware queues -- the runtime's default; under another deal of
    # queues, GPU_MAX_HW_QUEUES / DEBUG_HIP_DYNAMIC_QUEUES in the environment, only the results are asserted)
    if not (os.environ.get("GPU_MAX_HW_QUEUES") or os.environ.get("DEBUG_HIP_DYNAMIC_QUEUES")):
        assert many >= 2.2 * one, (one, many)


def test_mixed_entries_run_side_by_side_on_one_cascador(built, gpu, model_file):
    """Batches, ragged jobs, traces, submit/wait tickets and single frames from different threads at once: same
    results as alone; a pending ticket no longer blocks the other entry points."""
    import torch
    from jda_amd import api, synth
    p, _ = model_file((3, 70, 9, 5), 8, seed=71, cart_th=-0.9, norm_every=9)
    c = api.Cascador(p)
    batch = synth.make_frames(12, 240, 180, seed=1)
    imgs = [synth.make_frames(1, 100 + 13 * i, 90 + 7 * i, seed=2, first=i)[0] for i in range(9)]
    d_batch = torch.from_numpy(batch).to(gpu)
    want_batch = c.detect_batch(batch)
    want_rag = c.detect_ragged(imgs)
    want_tr = c.trace(batch[:2])
    errors = []

    def guard(fn):
        def run():
            try:
                for _ in range(6):
                    fn()
            except Exception as e:       # noqa: BLE001
                errors.append(repr(e))
        return run

    def do_batch():
        for a, b in zip(c.detect_batch(batch), want_batch):
            _eq(a, b, "batch")

    def do_ragged():
        for a, b in zip(c.detect_ragged(imgs), want_rag):
            _eq(a, b, "ragged")

    def do_trace():
        tr = c.trace(batch[:2])
        for k in want_tr:
            assert same(tr[k], want_tr[k]), k

    def do_single():
        for i in (0, 5, 11):
            _eq(c.detect(batch[i]), want_batch[i], "single")

    lock = threading.Lock()              # (api.Cascador keeps its tickets in one dict)

    def do_tickets():
        with lock:
            t1 = c.submit_batch_device(d_batch)
            t2 = c.submit_batch_host(batch)
        r1 = c.wait_batch(t1)
        r2 = c.wait_batch(t2)
        for a, b, w in zip(r1, r2, want_batch):
            _eq(a, w, "ticket dev"); _eq(b, w, "ticket host")

    ths = [threading.Thread(target=guard(f)) for f in (do_batch, do_ragged, do_trace, do_single, do_single, do_tickets)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    # a pending ticket and a synchronous call on the same cascador
    t = c.submit_batch_device(d_batch)
    for a, b in zip(c.detect_batch(batch), want_batch):
        _eq(a, b)
    for a, b in zip(c.wait_batch(t), want_batch):
        _eq(a, b)


def test_lane_pool_is_capped_and_callers_queue_up(built, gpu, model_file):
    """max_lanes = 2: eight threads on one cascador still all get their (correct) results -- callers beyond the cap
    wait for a lane instead of creating one -- and a big synchronous batch, which would take two lanes, runs on the one
    that is left."""
    from jda_amd import api, synth
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-1.0, norm_every=5)
    c = api.Cascador(p)
    c.set_option("max_lanes", 2)
    for key, bad in (("max_lanes", -1), ("workspace_mb", 0), ("handoff", -5)):           # refused, not applied
        with pytest.raises(api.JdaError):
            c.set_option(key, bad)
    assert c.get_option("max_lanes") == 2
    frames = synth.make_frames(8, 200, 150, seed=5)
    want = [c.detect(f) for f in frames]
    errors = []

    def work(t):
        try:
            for r in range(20):
                _eq(c.detect(frames[(t + r) % 8]), want[(t + r) % 8], (t, r))
        except Exception as e:           # noqa: BLE001
            errors.append(repr(e))
    ths = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    c.set_option("lanes_min_windows", 1)                   # every batch asks for two lanes now
    batch = synth.make_frames(6, 200, 150, seed=6)
    want_b = c.detect_batch(batch)
    t = c.submit_batch_host(batch)                           # holds one of the two lanes until its Wait
    for a, b in zip(c.detect_batch(batch), want_b):
        _eq(a, b, "batch next to a ticket")
    for a, b in zip(c.wait_batch(t), want_b):
        _eq(a, b, "ticket")
    c.close()


def test_release_drains_a_ticket_nobody_waited_for(built, gpu, model_file):
    """jdaCascadorRelease with a submitted batch still pending: the batch is drained (helper thread joined, stream
    synchronised), nothing crashes, and the device memory comes back."""
    import torch
    from jda_amd import api, synth
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-1.0, norm_every=5)
    batch = synth.make_frames(32, 320, 240, seed=7)
    d = torch.from_numpy(batch).to(gpu)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        c = api.Cascador(p)
        c.submit_batch_host(batch)
        c.submit_batch_device(d)
        c.close()                                            # no Wait
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < (64 << 20)


def test_idle_lanes_give_their_memory_back_while_others_call(built, gpu, model_file):
    """A burst of concurrent callers leaves lanes behind, each with a workspace; a lane that sits out `lane_idle_calls`
    hand-outs gives its buffers back.  The buffers are moved out under the cascador's mutex and freed AFTER it is
    released (hipFree synchronises the device: r04 freed under the lock) -- here while two threads keep calling:
    their results stay those of a quiet cascador and the device memory of the burst comes back."""
    import torch
    from jda_amd import api, synth
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-1.0, norm_every=5)
    c = api.Cascador(p)
    c.set_option("lane_idle_calls", 4)
    frames = synth.make_frames(8, 200, 150, seed=5)
    big = synth.make_frames(48, 320, 240, seed=6)            # a batch: its lane holds a workspace worth trimming
    want = [c.detect(f) for f in frames]
    want_big = c.detect_batch(big)
    errors = []

    def burst(t):
        try:
            for a, b in zip(c.detect_batch(big), want_big):
                _eq(a, b, ("burst", t))
        except Exception as e:           # noqa: BLE001
            errors.append(repr(e))
    ths = [threading.Thread(target=burst, args=(t,)) for t in range(6)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    torch.cuda.synchronize()
    free_burst = torch.cuda.mem_get_info()[0]

    def steady(t):
        try:
            for r in range(60):
                _eq(c.detect(frames[(t + r) % 8]), want[(t + r) % 8], (t, r))
        except Exception as e:           # noqa: BLE001
            errors.append(repr(e))
    ths = [threading.Thread(target=steady, args=(t,)) for t in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    torch.cuda.synchronize()
    free_after = torch.cuda.mem_get_info()[0]
    assert free_after > free_burst + (32 << 20), (free_burst, free_after)      # the idle lanes' workspaces are gone
    for a, b in zip(c.detect_batch(big), want_big):                             # ... and come back on demand
        _eq(a, b, "after the trim")
    c.close()


@pytest.mark.gpu
def test_bounded_queues_overflow_is_noticed_and_the_pass_rerun(built, model_file):
    """r06: the survivor queues of a pass are sized from the fractions earlier passes left in them (option ws_bound), not for
    every window.  A pass that outgrows them -- here forced: no floor, no safety factor, a cascade that keeps far more than
    the first-pass guess -- must be noticed through its counters and run again with room: same results as with worst-case
    queues, jdaStats.ws_regrows > 0, an empty error string; uniform batch, tickets, ragged job, dialect CPP."""
    import torch
    from jda_amd import api, synth
    from conftest import same
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-2.5, norm_every=5)      # (keeps a third of the windows and more)
    frames = synth.make_frames(6, 200, 150, seed=11)
    d = torch.from_numpy(frames).cuda()
    rng = np.random.default_rng(2)
    imgs = [np.ascontiguousarray(frames[i % 6][:int(rng.integers(60, 151)), :int(rng.integers(60, 201))]) for i in range(20)]
    c0 = api.Cascador(p)
    c0.set_option("ws_bound", 0)
    want = c0.detect_batch_device(d)
    want_rag = c0.detect_ragged(imgs)
    want_cpp = c0.detect_batch_cpp_device(d)
    want_rag_cpp = c0.detect_ragged_cpp(imgs)
    assert sum(len(w["scores"]) for w in want) > 0 and sum(len(w["scores"]) for w in want_cpp) > 0

    def fresh():
        c = api.Cascador(p)
        c.set_option("ws_min_entries", 1); c.set_option("ws_factor_pct", 100); c.set_option("device_post_min_frames", 2)
        c.set_option("dense", 0)              # (the sparse pipeline is what has queues; dense mode is the last case below)
        return c
    c = fresh()
    got, st = c.detect_batch_device(d, stats=True)
    assert st["ws_regrows"] >= 1 and api.last_error() == ""
    for a, b in zip(got, want):
        for k in ("bboxes", "scores", "shapes"):
            assert same(a[k], b[k]), k
    assert st["patch_n"] == st["scan_patch_n"]
    got2, st2 = c.detect_batch_device(d, stats=True)           # the rerun taught the plan its fractions
    for a, b in zip(got2, want):
        assert same(a["scores"], b["scores"])
    c = fresh()
    got = c.wait_batch(c.submit_batch_device(d))
    for a, b in zip(got, want):
        for k in ("bboxes", "scores", "shapes"):
            assert same(a[k], b[k]), k
    c = fresh()
    got, st = c.detect_ragged(imgs, stats=True)
    assert st["ws_regrows"] >= 1
    for a, b in zip(got, want_rag):
        for k in ("bboxes", "scores", "shapes"):
            assert same(a[k], b[k]), k
    c = fresh()
    got, st = c.detect_batch_cpp_device(d, stats=True)
    assert st["ws_regrows"] >= 1
    for a, b in zip(got, want_cpp):
        for k in ("rects", "scores", "shapes"):
            assert same(a[k], b[k]), k
    c = fresh()
    got, st = c.detect_ragged_cpp(imgs, stats=True)
    assert st["ws_regrows"] >= 1
    for a, b in zip(got, want_rag_cpp):
        for k in ("rects", "scores", "shapes"):
            assert same(a[k], b[k]), k
    # an all-pass model: the first pass finds most windows alive -> dense mode, which needs per-window state
    pa, _ = model_file((2, 8, 5, 3), 8, seed=5)
    ca = fresh(); ca.close()
    ca = api.Cascador(pa); ca.set_option("ws_min_entries", 1)
    c0 = api.Cascador(pa); c0.set_option("ws_bound", 0)
    ga, sa = ca.detect_batch_device(d, th=0.0, stats=True)
    wa = c0.detect_batch_device(d, th=0.0)
    assert sa["dense_passes"] >= 1
    for a, b in zip(ga, wa):
        for k in ("bboxes", "scores", "shapes"):
            assert same(a[k], b[k]), k


def test_lanes_get_hardware_queues_of_their_own(built, gpu, model_file, monkeypatch):
    """The runtime deals a process's streams to four hardware queues -- which one depends on every stream the host program
    created before -- and the kernels of one queue run one after the other (tools/experiments/hwq_probe.hip).  The
    cascador probes where its streams landed and hands them out by queue (host.h: StreamPool): however many streams the
    program made first, three lanes in flight have three queues.  hwq_place = 0 is the runtime's deal; results are the
    same either way."""
    import torch
    from jda_amd import api, synth
    p, _ = model_file((3, 20, 5, 4), 8, seed=3, cart_th=-1.0, norm_every=5)
    batch = synth.make_frames(6, 200, 150, seed=6)
    hw = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    outs = []
    for dummies in (0, 1, 2):
        keep = [torch.cuda.Stream() for _ in range(dummies)]           # the host program's own streams, used once
        for s_ in keep:
            with torch.cuda.stream(s_):
                torch.zeros(16, device="cuda").add_(1)
        torch.cuda.synchronize()
        c = api.Cascador(p)
        tickets = [c.submit_batch_host(batch) for _ in range(3)]       # three lanes held at once
        outs.append([c.wait_batch(t) for t in tickets])
        queues, streams, probes, worst = (c.get_option(k) for k in ("hwq_queues", "hwq_streams", "hwq_probes", "hwq_max_mains"))
        assert queues >= min(3, hw), (dummies, queues)
        assert worst == 1 or hw < 3, (dummies, "two lanes share a hardware queue", queues, streams, probes)
        assert probes >= 2 and streams <= 24, (streams, probes)
        c.close()
    monkeypatch.setenv("JDA_HWQ_PLACE", "0")
    c = api.Cascador(p)
    tickets = [c.submit_batch_host(batch) for _ in range(3)]
    plain = [c.wait_batch(t) for t in tickets]
    assert c.get_option("hwq_queues") == 0 and c.get_option("hwq_probes") == 0
    c.close()
    for o in outs:
        for got, want in zip(o, plain):
            for a, b in zip(got, want):
                _eq(a, b, "placed vs plain streams")
