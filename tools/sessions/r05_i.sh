#!/bin/bash
TAG=${1:-r05_i}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python tools/pipe_variants.py "" "JDA_SCAN_P_WGS=2 JDA_SCAN_P_BLOCK=384" "JDA_SCAN_P_WGS=2 JDA_SCAN_P_BLOCK=512" \
  "JDA_SCAN_P_MIN_SLOTS=3" "JDA_SCAN_P_MIN_SLOTS=2" "JDA_SCAN_P_MIN_SLOTS=3 JDA_SCAN_P_TILE_KB=24" "JDA_SCAN_P_BLOCK=832" "JDA_FIN_GRID_DIV=2" "JDA_FIN_GRID_DIV=8" "" > $O/pipe_variants.txt 2>&1
cat $O/pipe_variants.txt
