#!/bin/bash
# Round 5: kernel timelines of the submit/wait pipeline with and without the chain of scans
TAG=${1:-r05_e}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PIPE_AHEAD=2 PIPE_STEPS=40
for v in 1 0; do
  JDA_SCAN_CHAIN=$v timeout 300 rocprofv3 --kernel-trace -d $O/kt$v -- python $R/tools/pipe.py > $O/run$v.txt 2>&1
  python $R/tools/bench_timeline.py $(find $O/kt$v -name "*.db" | head -1) > $O/timeline_chain$v.txt 2>&1
  rm -rf $O/kt$v
done
grep submit $O/run1.txt $O/run0.txt
head -60 $O/timeline_chain1.txt
