#!/bin/bash
# round 4: per-task clocks of k_scan_p (timing build) and per-level kernel times, old against new
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export JDA_LANES=1 JDA_SIDE_STREAM=0
for v in "JDA_SCAN_P_LG=64444" "JDA_SCAN_P_LG=66666" "JDA_SCAN_P_LG=64444 JDA_SCAN_P_SLOTS=3"; do
  echo "=== $v"
  env $v JDA_LIB_PATH=jda_amd/libjda_timing.so JDA_SCAN_P=1 JDA_NO_GLOBAL_SCAN=1 timeout 300 python tools/scan_p_timing.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04_b_timing.log 2>&1
cat gpurun_out/r04_b_timing.log
