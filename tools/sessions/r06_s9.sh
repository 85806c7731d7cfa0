#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s9; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ragged.py tests/test_fddb.py tests/test_abi.py tests/test_device_post.py -x -q -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for r in 0 1 2 3 4 5 6 7; do timeout 120 python tools/shard_job.py 20 8 $r 2>&1 | tail -1; done | tee $O/shards.txt
SHARD_BY_COUNT=1 timeout 120 python tools/shard_job.py 20 8 0 2>&1 | tail -1
python tools/fddb_job.py 10 "" "" 2>&1 | grep "per job"
tools/experiments/lds_occ.bin 2>&1 | tee gpurun_out/r06_s9/lds_occ.txt
