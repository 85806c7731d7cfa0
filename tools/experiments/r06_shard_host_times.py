import os, sys, time
sys.path.insert(0, ".")
exec(open("./tools/shard_job.py").read().split("for _ in range(3):")[0])
for _ in range(3): job()
import ctypes as C
t0=time.perf_counter(); job(); t1=time.perf_counter()
print("python-side total %.3f ms" % ((t1-t0)*1e3))
c.set_option("debug_times", 1)
t0=time.perf_counter(); job(); t1=time.perf_counter()
print("python-side total with prints %.3f ms" % ((t1-t0)*1e3))
