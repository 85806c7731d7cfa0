#!/bin/bash
# Kernel + memory-copy timeline of the host-frame stream (jdaDetectBatchSubmitHost / Wait), pinned and pageable sources.
cd "$(dirname "$0")/.." && R=$PWD && export TMPDIR=/tmp
mkdir -p gpurun_out
for kind in ${@:-pinned pageable}; do
  rm -rf /tmp/ht && mkdir -p /tmp/ht
  (cd /tmp && HOST_ONLY=$kind HOST_STEPS=12 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/ht -o ht -- python $R/tools/host_variants.py "" 2>&1 | grep -v amdgpu | grep defaults)
  DB=$(find /tmp/ht -name "*.db" | head -1)
  python tools/host_timeline.py $DB > gpurun_out/host_timeline_$kind.txt 2> /dev/null
done
