#!/bin/bash
# round 4, first GPU session of k_scan_p: parity against k_scan, then step / scan times under several settings
set -x
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export JDA_LANES=1 JDA_SIDE_STREAM=0
timeout 900 python tools/scan_p_check.py > gpurun_out/r04_a_check.log 2>&1; echo "check rc $?" >> gpurun_out/r04_a_check.log
tail -5 gpurun_out/r04_a_check.log
timeout 600 python tools/variants.py "" "JDA_SCAN_P=1" "JDA_SCAN_P=1 JDA_SCAN_P_BLOCK=512" "JDA_SCAN_P=1 JDA_SCAN_P_BLOCK=512 JDA_SCAN_P_WGS=2" \
  "JDA_SCAN_P=1 JDA_SCAN_P_BLOCK=768" "JDA_SCAN_P=1 JDA_SCAN_P_LG=66666" "JDA_SCAN_P=1 JDA_SCAN_P_LG=65555" "JDA_SCAN_P=1 JDA_SCAN_P_LG=66444" \
  "JDA_SCAN_P=1 JDA_SCAN_P_OPTS=1" "JDA_SCAN_P=1 JDA_SCAN_P_OPTS=3" "JDA_SCAN_P=1 JDA_SCAN_P_B3=0" "JDA_SCAN_P=1 JDA_SCAN_P_B0=8 JDA_SCAN_P_B1=16 JDA_SCAN_P_B2=32 JDA_SCAN_P_B3=64 JDA_SCAN_P_B4=96 JDA_SCAN_P_LG=666444" \
  "JDA_SCAN_P=1 JDA_SCAN_P_SLOTS=3" "JDA_SCAN_P=1 JDA_SCAN_P_SLOTS=2" > gpurun_out/r04_a_variants.log 2>&1
cat gpurun_out/r04_a_variants.log
JDA_LIB_PATH=jda_amd/libjda_timing.so JDA_SCAN_P=1 timeout 300 python tools/scan_p_timing.py > gpurun_out/r04_a_timing.log 2>&1
cat gpurun_out/r04_a_timing.log
JDA_LIB_PATH=jda_amd/libjda_timing.so JDA_SCAN_P=1 JDA_SCAN_P_LG=66666 timeout 300 python tools/scan_p_timing.py > gpurun_out/r04_a_timing_uni.log 2>&1
cat gpurun_out/r04_a_timing_uni.log
