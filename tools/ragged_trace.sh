#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-ragged_trace}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/ragged_bench.py --variants device --reps 3 > $O/run.txt 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) k_ > $O/kernel_trace.txt
find $O -name "*.db" -delete; rm -rf $O/kt
grep -v amdgpu $O/run.txt | tail -2 | cut -c1-400
head -14 $O/kernel_trace.txt | cut -c1-210
grep "us grid" $O/kernel_trace.txt | tail -36 | cut -c1-120
