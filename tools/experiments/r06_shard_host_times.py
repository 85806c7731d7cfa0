"""Where a shard job's host time goes: the library's own marks (debug_times = 2: one line per job) and the Python side
(marshalling before the call, result collection after it).  usage: python tools/experiments/r06_shard_host_times.py"""
import os, sys, time
sys.path.insert(0, ".")
exec(open("./tools/shard_job.py").read().split("for _ in range(3):")[0])
for _ in range(3): job()
import torch
ts = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); job(); ts.append((time.perf_counter() - t0) * 1e3)
print("python-side job: median %.3f min %.3f ms" % (sorted(ts)[10], min(ts)))
c.set_option("debug_times", 2)
for _ in range(6):
    t0 = time.perf_counter(); job(); t1 = time.perf_counter()
    print("python-side total %.3f ms" % ((t1 - t0) * 1e3))
