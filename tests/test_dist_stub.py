"""jda_amd/csrc/dist.cpp with SEVERAL ranks on the one GPU of the test box.

RCCL refuses two ranks on one device, and the multi-GPU box is the driver's, not ours: so far libjda_dist.so had
only ever run in a group of one (tests/test_gpu_parity.py), its offsets and per-rank branches for rank >= 1
untested until the scaling run.  Here the SHIPPED libjda_dist.so runs as 2, 5 and 8 processes on device 0 with
its eight RCCL entry points interposed (LD_PRELOAD) by tests/c/rccl_stub.cpp, which moves the same bytes through
shared memory.  What is under test is everything dist.cpp does around those calls: blocks and counts of the
pipelined gather, who takes the exact path and when, receive offsets on rank 0, empty ranks, two gathers in
flight with the fallback between them, the 2-D copy of the counts on ranks other than 0, the grow-only buffers.
RCCL itself is covered where it can run (group of one here, the bench's first-contact self-test at N>1).
"""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_SRC = os.path.join(ROOT, "tests", "c", "rccl_stub.cpp")
STUB_SO = os.path.join(ROOT, "tests", "c", "librccl_stub.so")

WORKER = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from jda_amd import dist as jd

rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
WIDTH, BLOCK = 141, 48
if rank == 0:
    uid = jd.unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 60:
            raise SystemExit("no id from rank 0")
        time.sleep(0.01)
    uid = open(idfile, "rb").read()
assert bytes(uid).startswith(b"/jda_rccl_stub_"), "the stub is not the one answering: %%r" %% bytes(uid)[:24]
g = jd.CGather(rank, world, uid, 0, WIDTH, BLOCK)

def rows_of(job, r):
    rng = np.random.default_rng(1000 * job + r)
    kind = job %% 5
    if kind == 0:   n = int(rng.integers(0, BLOCK + 1))                      # everything fits
    elif kind == 1: n = 0 if r %% 2 else int(rng.integers(1, BLOCK))          # empty ranks in between
    elif kind == 2: n = int(rng.integers(BLOCK + 1, 4 * BLOCK)) if r == job %% world else int(rng.integers(0, 5))   # one rank overflows
    elif kind == 3: n = int(rng.integers(BLOCK + 1, 3 * BLOCK))              # all of them overflow
    else:           n = 0                                                   # nobody has anything
    m = rng.integers(-2**20, 2**20, size=(n, WIDTH)).astype(np.float32)
    if n: m[:, 0] = r
    return m

def want(job):
    return np.concatenate([rows_of(job, r) for r in range(world)])

def check(job, got, what):
    if rank != 0:
        assert got is None or len(got) == 0, (what, job, "rows on rank", rank)
        return
    w = want(job)
    assert got is not None and got.shape == w.shape and np.array_equal(got, w), (what, job, None if got is None else got.shape, w.shape)

# the first-contact self-test of bench.py, as it would run at this world size
assert jd.gather_selftest(g, rank, world, WIDTH, BLOCK) == "ok"
# pipelined: start() hands back the previous job's rows, two in flight at the seams
prev = None
for job in range(15):
    got = g.start(rows_of(job, rank))
    if prev is not None:
        check(prev, got, "start")
    prev = job
check(prev, g.drain(), "drain")
# two really in flight, every pairing of fitting / overflowing
for a, b in ((0, 2), (2, 0), (3, 2), (2, 3), (4, 3), (1, 4)):
    g.start_raw(rows_of(a, rank)); g.start_raw(rows_of(b, rank))
    check(a, g.collect(), "collect-1"); check(b, g.collect(), "collect-2")
# the blocking exact gather, sizes growing and shrinking (grow-only buffers)
for job in (3, 0, 2, 4, 3, 1):
    check(job, g.gather(rows_of(job, rank)), "exact")
# misuse is an error on every rank, not a hang
g.start_raw(rows_of(0, rank)); g.start_raw(rows_of(1, rank))
try:
    g.start_raw(rows_of(0, rank)); raise SystemExit("a third start was accepted")
except RuntimeError as e:
    assert "two gathers" in str(e)
try:
    g.gather(rows_of(0, rank)); raise SystemExit("a blocking gather overtook pending ones")
except RuntimeError as e:
    assert "pending" in str(e)
check(0, g.collect(), "after-misuse-1"); check(1, g.collect(), "after-misuse-2")
g.close()
print("rank %%d of %%d ok" %% (rank, world))
"""


def build_stub():
    if os.path.exists(STUB_SO) and os.path.getmtime(STUB_SO) >= os.path.getmtime(STUB_SRC):
        return STUB_SO
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc to build the RCCL stand-in")
    subprocess.run([hipcc, "-O1", "-fPIC", "-shared", "-w", STUB_SRC, "-o", STUB_SO, "-lpthread", "-lrt"], check=True, timeout=300)
    return STUB_SO


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 5, 8])
def test_dist_library_with_several_ranks_on_one_gpu(world):
    stub = build_stub()
    env = dict(os.environ, LD_PRELOAD=stub, PYTHONPATH=ROOT)
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "worker.py")
        with open(script, "w") as f:
            f.write(WORKER % {"root": ROOT})
        idfile = os.path.join(tmp, "id")
        procs = [subprocess.Popen([sys.executable, script, str(r), str(world), idfile], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs, bad = [], False
        for r, p in enumerate(procs):
            try:
                out, _ = p.communicate(timeout=240)
            except subprocess.TimeoutExpired:
                bad = True
                for q in procs:
                    if q.poll() is None:
                        q.kill()
                out, _ = p.communicate()
                out = (out or "") + "\n[timeout]"
            outs.append(out)
            bad = bad or p.returncode != 0 or ("rank %d of %d ok" % (r, world)) not in out
        assert not bad, "\n".join("--- rank %d\n%s" % (r, o[-2000:]) for r, o in enumerate(outs))


@pytest.mark.gpu
def test_bench_two_ranks_with_the_c_gather_in_the_loop():
    """bench.py's N>1 flow as the driver's scaling run takes it -- libjda_dist.so set up from ids broadcast by rank 0,
    the first-contact self-test, a pipelined gather per step, the FDDB shard gather -- with two ranks on this GPU:
    torch's own group over gloo, the library's RCCL calls answered by the stand-in."""
    import json
    stub = build_stub()
    env = dict(os.environ, LD_PRELOAD=stub, JDA_BENCH_BACKEND="gloo", JDA_BENCH_ONE_GPU="1", JDA_BENCH_C_GATHER_ON_GLOO="1", JDA_DENSE="1")
    env.pop("JDA_BENCH_GATHER", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32",
           "--no-allpass", "--no-config2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert cfg["gather"].startswith("libjda_dist.so"), cfg["gather"]
    assert cfg["dist_selftest"].startswith("ok"), cfg["dist_selftest"]
    assert d["regimes"]["cascade"]["detections_after_nms"] > 0 and cfg["fddb_images_per_s"] > 0
    assert "falling back" not in out.stderr, out.stderr[-2000:]
    # r06: an N-rank line carries what the one-GPU line carries -- the CPU baseline (rank 0, after the timed regions), the
    # parity verdict on the last timed call of every leg, the dialect-CPP legs (the FDDB one sharded over the ranks)
    assert d["cpu_baseline"] and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] in ("reference", "port"), d["cpu_baseline"]
    assert cfg["parity_checked"] in ("reference", "port"), (cfg["parity_checked"], d.get("parity"))
    assert all(cfg["parity_legs"].values()) and {"headline", "fddb", "cpp", "fddb_cpp"} <= set(cfg["parity_legs"]), cfg["parity_legs"]
    assert cfg["cpp_windows_per_s"] > 0 and cfg["fddb_cpp_images_per_s"] > 0 and cfg["host_frames_windows_per_s"] > 0
