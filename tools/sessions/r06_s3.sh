#!/bin/bash
# CPP job: per-dispatch rows (grid, LDS bytes, duration) of one job on one lane; + shard host times after the table rework
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
JDA_RAGGED_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/cpp_job.py 1 > $O/run.txt 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) k_ > $O/cpp_job_dispatches.txt
rm -rf $O/kt
python tools/experiments/r06_shard_host_times.py 2>&1 | grep -v amdgpu > $O/shard_host.txt
for r in 0 3 7; do timeout 120 python tools/shard_job.py 20 8 $r 2>&1 | tail -1; done > $O/shard_jobs.txt
cat $O/shard_host.txt $O/shard_jobs.txt
