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
cd /tmp && export TMPDIR=/tmp
for v in "JDA_SCAN_P=0" "JDA_SCAN_P=1" "JDA_SCAN_P=1 JDA_SCAN_P_BLOCK=512 JDA_SCAN_P_WGS=2"; do
  rm -rf /tmp/kt; 
  env $v VAR_STEPS=5 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $GRAFT_REPO_ROOT/tools/variants.py "" > /tmp/kt.log 2>&1
  echo "=== $v"; tail -2 /tmp/kt.log
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) k_scan 2>&1 | grep -i "k_scan" | cut -c1-150 | head -30
done > $GRAFT_REPO_ROOT/gpurun_out/r04_b_kt.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r04_b_kt.log
