#!/bin/bash
TAG=${1:-r05_h}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; timeout 300 python tools/fddb_job.py 5 "" "JDA_RAGGED_CHUNK_WINDOWS=6000000" "JDA_RAGGED_CHUNK_WINDOWS=3000000" "JDA_DEVICE_POST=0" > $O/job.txt 2>&1; cat $O/job.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt -- python $R/tools/fddb_job.py 3 > $O/run.txt 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python $R/tools/host_timeline.py $DB > $O/timeline_all.txt 2>/dev/null
python $R/tools/rocpd_summary.py $DB k_ > $O/kernel_trace.txt
rm -rf $O/kt
tail -120 $O/timeline_all.txt > $O/timeline_tail.txt
head -20 $O/kernel_trace.txt | cut -c1-200
