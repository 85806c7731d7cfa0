#!/bin/bash
# kernel / copy timeline of a 356-image ragged job (one rank's shard of the FDDB-sized job at 8 GPUs), device-resident images
cd "$(dirname "$0")/.." && R=$PWD && export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf /tmp/tl && mkdir -p /tmp/tl
(cd /tmp && env $1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o tl -- python $R/tools/ragged_bench.py --images 356 --variants device --reps 5 2>/dev/null | grep "^{" | cut -c1-330)
python tools/host_timeline.py $(find /tmp/tl -name "*.db" | head -1) > gpurun_out/ragged_shard_timeline.txt 2>&1
tail -75 gpurun_out/ragged_shard_timeline.txt
