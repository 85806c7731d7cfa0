"""bench.py's command line without a GPU: `--gpus N` with no launcher in the environment must turn into one
torch.distributed.run job of N ranks on 127.0.0.1 (the driver runs `python bench.py --gpus N --steps K --warmup W`)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_self_spawn_argv_is_one_rank_per_gpu_on_localhost():
    import bench
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "3"]
    cmd = bench.self_spawn_argv(argv, 8, 29512)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29512"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                        # the ranks see the same arguments, --gpus included


def test_self_spawn_only_without_a_launcher(monkeypatch):
    import bench
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    one = argparse.Namespace(gpus=1)
    assert bench.maybe_self_spawn(one, []) is None                      # N = 1: this process is the job
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.maybe_self_spawn(argparse.Namespace(gpus=4), []) is None   # already a rank of a launcher's job
    monkeypatch.delenv("RANK"); monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.delenv("JDA_BENCH_ONE_GPU", raising=False)
    # this container has no GPU: asking for 2 is refused loudly instead of running one rank and printing n_gpus 1
    import torch
    if not torch.cuda.is_available():
        assert bench.maybe_self_spawn(argparse.Namespace(gpus=2), ["--gpus", "2"]) == 2


def test_free_port_is_bindable():
    import socket
    import bench
    p = bench.free_port()
    assert 1024 < p < 65536
    s = socket.socket(); s.bind(("127.0.0.1", p)); s.close()
