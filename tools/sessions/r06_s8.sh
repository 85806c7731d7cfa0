#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s8; mkdir -p $O; cd $R
export JDA_EXP_A=1
timeout 900 python -m pytest tests/test_ragged.py tests/test_fddb.py tests/test_cpp_entries.py tests/test_device_post.py tests/test_reentrant.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for r in 0 3 7; do timeout 120 python tools/shard_job.py 20 8 $r 2>&1 | tail -1; done
python tools/experiments/r06_shard_host_times.py 2>&1 | grep -v amdgpu | tail -9
timeout 100 python tools/fddb_job.py 2>&1 | tail -3
