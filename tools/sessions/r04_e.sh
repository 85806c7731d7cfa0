#!/bin/bash
# round 4, session 2: parity of the new k_scan_p options (own tile cut, whole stage 0 with direct mid queue, pair tasks of 8 / 4),
# serialised step times per variant ($VARIANTS, ';'-separated), per-task clocks ($TIMING), headline A/B ($BENCH)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TAG=${TAG:-r04_e}
timeout 900 python tools/scan_p_check.py ${CHECK:-quick} > gpurun_out/${TAG}_check.log 2>&1; echo "check rc $?" >> gpurun_out/${TAG}_check.log
grep -c " ok stats ok" gpurun_out/${TAG}_check.log; grep -i "mismatch\|error\|Traceback" gpurun_out/${TAG}_check.log | head; tail -2 gpurun_out/${TAG}_check.log
IFS=';' read -ra VS <<< "${VARIANTS:-JDA_SCAN_P=0;}"
JDA_LANES=1 JDA_SIDE_STREAM=0 timeout 900 python tools/variants.py "${VS[@]}" 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_variants.log
cat gpurun_out/${TAG}_variants.log
if [ -n "$TIMING" ]; then
IFS=';' read -ra TS <<< "$TIMING"
for v in "${TS[@]}"; do
  echo "=== $v"
  env $v JDA_LANES=1 JDA_SIDE_STREAM=0 JDA_LIB_PATH=jda_amd/libjda_timing.so JDA_NO_GLOBAL_SCAN=1 timeout 300 python tools/scan_p_timing.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/${TAG}_timing.log 2>&1
cat gpurun_out/${TAG}_timing.log
fi
if [ -n "$BENCH" ]; then
IFS=';' read -ra BS <<< "$BENCH"
TAG=${TAG}_bench bash tools/sessions/r04_bench_ab.sh "${BS[@]}"
fi
