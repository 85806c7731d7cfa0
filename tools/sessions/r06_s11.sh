#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s11; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
JDA_RAGGED_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/cpp_job.py 3 > $O/run.txt 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) k_ > $O/cpp_job_dispatches.txt
rm -rf $O/kt
grep "CPP ragged" $O/run.txt
head -12 $O/cpp_job_dispatches.txt | cut -c1-150
grep -n "k_scan<double" $O/cpp_job_dispatches.txt | sed -n 20,30p | cut -c1-120
for i in 1 2 3; do python tools/cpp_job.py 5 2>&1 | grep "CPP ragged"; done
