#!/bin/bash
# round 4: k_scan_p iteration -- quick parity, step times under settings ($VARIANTS, ';'-separated), per-task clocks ($TIMING)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export JDA_LANES=1 JDA_SIDE_STREAM=0
TAG=${TAG:-r04_d}
timeout 600 python tools/scan_p_check.py ${CHECK:-quick} > gpurun_out/${TAG}_check.log 2>&1; echo "check rc $?" >> gpurun_out/${TAG}_check.log
grep -c " ok stats ok" gpurun_out/${TAG}_check.log; grep -i "mismatch\|error\|Traceback" gpurun_out/${TAG}_check.log | head; tail -2 gpurun_out/${TAG}_check.log
IFS=';' read -ra VS <<< "${VARIANTS:-JDA_SCAN_P=0;JDA_SCAN_P=1}"
timeout 900 python tools/variants.py "${VS[@]}" 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_variants.log
cat gpurun_out/${TAG}_variants.log
IFS=';' read -ra TS <<< "${TIMING:-JDA_SCAN_P_LG=64444}"
for v in "${TS[@]}"; do
  echo "=== $v"
  env $v JDA_LIB_PATH=jda_amd/libjda_timing.so JDA_SCAN_P=1 JDA_NO_GLOBAL_SCAN=1 timeout 300 python tools/scan_p_timing.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/${TAG}_timing.log 2>&1
cat gpurun_out/${TAG}_timing.log
