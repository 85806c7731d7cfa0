#!/bin/bash
# per-level kernel times (rocprofv3 kernel trace) of one synchronous step x5 under the given environments
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r04_kt}
cd /tmp; export TMPDIR=/tmp
export JDA_LANES=1 JDA_SIDE_STREAM=0 VAR_STEPS=5
for v in "$@"; do
  rm -rf /tmp/kt
  env $v timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -- python $R/tools/variants.py "" > /tmp/kt.log 2>&1
  echo "=== $v"; grep "step" /tmp/kt.log | cut -c1-160
  python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) k_scan 2>&1 | grep "us grid" | tail -5 | cut -c1-140
done > $R/gpurun_out/$TAG.log 2>&1
cat $R/gpurun_out/$TAG.log
