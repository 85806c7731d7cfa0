#!/bin/bash
# kernel timeline of BASELINE configs[2] (256 x 1080p) in submit/wait mode: where the pipeline waits
cd "$(dirname "$0")/.." && R=$PWD && export TMPDIR=/tmp
mkdir -p gpurun_out; TAG=${TAG:-r04_c2tl}
i=0
for v in "$@"; do
  i=$((i+1)); rm -rf /tmp/tl && mkdir -p /tmp/tl
  (cd /tmp && env $v timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o tl -- python $R/tools/config2.py pipe 6 2>/dev/null | grep "^{" | cut -c1-400)
  DB=$(find /tmp/tl -name "*.db" | head -1)
  { echo "=== $v"; python tools/host_timeline.py $DB; } > gpurun_out/${TAG}_$i.txt 2>&1
  tail -60 gpurun_out/${TAG}_$i.txt | head -50
done
