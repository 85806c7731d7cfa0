#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s5; mkdir -p $O; cd $R
run() { echo "== $*"; for r in 0 3 7; do env "$@" timeout 120 python tools/shard_job.py 20 8 $r 2>&1 | tail -1; done; }
run A=0 > $O/shards.txt 2>&1
run JDA_EXP_A=1 >> $O/shards.txt 2>&1
run A=0 >> $O/shards.txt 2>&1
run JDA_EXP_A=1 >> $O/shards.txt 2>&1
cat $O/shards.txt
JDA_EXP_A=1 timeout 300 python -m pytest tests/test_ragged.py -x -q -m gpu 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
JDA_EXP_A=1 timeout 200 rocprofv3 --kernel-trace -d $O/kt -- python $R/tools/shard_job.py 10 8 0 > $O/run.txt 2>&1
cd $R
python tools/shard_timeline.py $(find $O/kt -name "*.db" | head -1) 2>&1 | head -16
rm -rf $O/kt
