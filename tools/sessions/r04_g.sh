#!/bin/bash
# round 4: tile shapes chosen for LDS bank conflicts (JDA_TILES is read once per process: one process per variant)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; TAG=${TAG:-r04_g}
for v in "$@"; do
  env $v JDA_LANES=1 JDA_SIDE_STREAM=0 timeout 300 python tools/variants.py "$v" 2>&1 | grep -v amdgpu.ids
done > gpurun_out/$TAG.log 2>&1
cat gpurun_out/$TAG.log
TAG=${TAG}_bench bash tools/sessions/r04_bench_ab.sh "$@"
