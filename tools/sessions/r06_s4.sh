#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s4; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ragged.py tests/test_fddb.py tests/test_cpp_entries.py tests/test_device_post.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
run() { echo "== $*"; for r in 0 3 7; do env "$@" timeout 120 python tools/shard_job.py 20 8 $r 2>&1 | tail -1; done; }
run A=0 > $O/shards.txt 2>&1
run JDA_RAGGED_MERGE=1 >> $O/shards.txt 2>&1
run JDA_RAGGED_MERGE=1 JDA_RAGGED_SIDE=0 >> $O/shards.txt 2>&1
run JDA_SCAN_P_RAGGED=0 >> $O/shards.txt 2>&1
run JDA_SCAN_P_RAGGED=0 JDA_RAGGED_MERGE=1 >> $O/shards.txt 2>&1
run A=0 >> $O/shards.txt 2>&1
cat $O/shards.txt
