#!/bin/bash
# Round 5: the chain of scans (scan_chain) in the submit/wait pipeline, with the co-residency knobs around it
TAG=${1:-r05_d}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python tools/pipe_variants.py "JDA_SCAN_CHAIN=0" "" "JDA_SCAN_CHAIN=0" "" \
  "JDA_SCAN_P_SLOTS=4" "JDA_SCAN_P_SLOTS=6" "JDA_SCAN_P_SLOTS=0" "JDA_FIN_TILE=0" "JDA_SCAN_P_SLOTS=4 JDA_FIN_TILE=0" \
  "JDA_SCAN_P_LDS_KB=128" "JDA_SCAN_P_LDS_KB=112" "JDA_SIDE_STREAM=0" "JDA_SCAN_CHAIN=2" > $O/pipe_variants.txt 2>&1
cat $O/pipe_variants.txt
export VAR_STEPS=10
timeout 300 python tools/variants.py "JDA_SCAN_CHAIN=0" "" "JDA_SCAN_CHAIN=2" > $O/sync_variants.txt 2>&1; cat $O/sync_variants.txt
