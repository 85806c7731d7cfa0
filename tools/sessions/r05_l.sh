#!/bin/bash
TAG=${1:-r05_l}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python tools/pipe_variants.py "" "JDA_FIN_GRID_DIV=2" "" "JDA_FIN_GRID_DIV=2" "JDA_FIN_GRID_DIV=1" "JDA_FIN_GRID_DIV=3" "JDA_FIN_GRID_DIV=16" "" "JDA_FIN_GRID_DIV=2" > $O/pipe_variants.txt 2>&1
cat $O/pipe_variants.txt
