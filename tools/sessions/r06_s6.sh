#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s6; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in "JDA_SIDE_STREAM=0" "JDA_SIDE_STREAM=0 JDA_SCAN_P_RAGGED=0"; do
env JDA_EXP_A=1 $v timeout 200 rocprofv3 --kernel-trace -d $O/kt -- python $R/tools/shard_job.py 6 8 0 > $O/run.txt 2>&1
echo "== $v"; python $R/tools/shard_timeline.py $(find $O/kt -name "*.db" | head -1) 2>&1 | head -12
rm -rf $O/kt
done
