#!/bin/bash
TAG=${1:-r05_o}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for n in 356 711 1422; do echo "== $n images"; FDDB_N=$n timeout 300 python tools/fddb_job.py 20 "JDA_SCAN_P_RAGGED=0" "" "JDA_SCAN_P_RAGGED=0" "" 2>/dev/null; done > $O/shards.txt 2>&1
cat $O/shards.txt
