#!/bin/bash
# kernel trace of tools/variants.py under the given environment: tools/kt_variants.sh "JDA_X=1 JDA_Y=2"
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/ktv; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
env JDA_LANES=1 JDA_SIDE_STREAM=0 VAR_STEPS=10 $1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/variants.py "" > $O/run.txt 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) k_ | head -9 | cut -c1-150
find $O -name "*.db" -delete; rm -rf $O/kt
