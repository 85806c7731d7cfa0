#!/bin/bash
TAG=${1:-r05_f}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python tools/pipe_variants.py "JDA_SCAN_CHAIN=0" "" "JDA_SCAN_PRIO=1" "JDA_SCAN_PRIO=1 JDA_SCAN_P_SLOTS=4" "JDA_SCAN_PRIO=1 JDA_FIN_TILE=0" \
  "JDA_SCAN_PRIO=1 JDA_SCAN_P_SLOTS=4 JDA_FIN_TILE=0" "JDA_SCAN_PRIO=1 JDA_SCAN_P_LDS_KB=112" "JDA_SCAN_PRIO=1 JDA_SIDE_STREAM=0" \
  "JDA_SCAN_CHAIN=0 JDA_SCAN_PRIO=1" "JDA_SCAN_CHAIN=0" > $O/pipe_variants.txt 2>&1
cat $O/pipe_variants.txt
