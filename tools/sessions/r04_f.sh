#!/bin/bash
# round 4: weight rows on their own 128-byte lines (JDA_W_PAD): parity, serialised step, bench headline, configs[4] all-pass
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; TAG=${TAG:-r04_f}
{
timeout 600 python tools/scan_p_check.py quick 2>&1 | tail -2
JDA_LANES=1 JDA_SIDE_STREAM=0 timeout 600 python tools/variants.py "" "JDA_W_PAD=0" 2>&1 | grep -v amdgpu.ids
for v in "JDA_W_PAD=1" "JDA_W_PAD=0"; do echo "=== $v"; env $v timeout 300 python tools/x_allpass.py --frames 1 --steps 2 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/$TAG.log 2>&1
cat gpurun_out/$TAG.log
TAG=${TAG}_bench bash tools/sessions/r04_bench_ab.sh "" "JDA_W_PAD=0"
