#!/bin/bash
# Round 5: kernel trace of configs[2] (256 x 1080p, scale 1.5) on the final code, every launch back to back on one stream
TAG=${1:-r05_s}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp VAR_STEPS=3
JDA_LANES=1 JDA_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/config2_variants.py "" > $O/run.txt 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) k_ > $O/config2_kernel_trace_stats.txt
rm -rf $O/kt; grep -v amdgpu $O/run.txt | tail -2; head -12 $O/config2_kernel_trace_stats.txt | cut -c1-180
