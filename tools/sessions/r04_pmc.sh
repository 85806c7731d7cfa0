#!/bin/bash
# round 4: SQ counters of the LDS-tiled scan, k_scan against k_scan_p (one synchronous 256-frame step x 8, serialised launches)
#   gpurun -- 'TAG=r04_pmc bash tools/sessions/r04_pmc.sh "JDA_SCAN_P=0" "JDA_SCAN_P=1"'
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r04_pmc}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export JDA_LANES=1 JDA_SIDE_STREAM=0 VAR_STEPS=5
i=0
for v in "$@"; do
  i=$((i+1))
  rm -rf /tmp/p1 /tmp/p2
  env $v timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/p1 -- python $R/tools/variants.py "" > /dev/null 2>&1
  env $v timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY -d /tmp/p2 -- python $R/tools/variants.py "" > /dev/null 2>&1
  { echo "=== $v"
    python $R/tools/rocpd_pmc.py $(find /tmp/p1 -name "*.db" | head -1) k_scan
    python $R/tools/rocpd_pmc.py $(find /tmp/p2 -name "*.db" | head -1) k_scan
  } | cut -c1-170 > $O/v$i.txt 2>&1
  cat $O/v$i.txt
done
