#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/rvu; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
JDA_LANES=1 JDA_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/ragged_vs_uniform.py 12000000 > $O/run.txt 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) k_ > $O/kernel_trace.txt
find $O -name "*.db" -delete; rm -rf $O/kt
grep "us grid" $O/kernel_trace.txt | tail -16 | cut -c1-125
