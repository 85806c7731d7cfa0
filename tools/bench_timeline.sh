#!/bin/bash
# Kernel timeline of the bench's headline leg (submit/wait, two batches in flight) in steady state: which kernels of
# the two tickets really overlap.   gpurun -- 'TAG=r04_tl bash tools/bench_timeline.sh "" "JDA_SCAN_P_SLOTS=5"'
cd "$(dirname "$0")/.." && R=$PWD && export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${TAG:-r04_tl}
i=0
for v in "$@"; do
  i=$((i+1))
  rm -rf /tmp/tl && mkdir -p /tmp/tl
  (cd /tmp && env $v timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $R/tools/pipe.py 2>/dev/null | tail -2)
  DB=$(find /tmp/tl -name "*.db" | head -1)
  { echo "=== $v"; python tools/bench_timeline.py $DB; } > gpurun_out/${TAG}_$i.txt 2>&1
  head -5 gpurun_out/${TAG}_$i.txt
done
