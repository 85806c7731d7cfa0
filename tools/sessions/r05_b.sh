#!/bin/bash
# Round 5: where do the big-window levels belong?  lds_win_max sweeps on configs[1] and configs[2], single-frame latency.
TAG=${1:-r05_b}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export VAR_STEPS=10
timeout 300 python tools/variants.py "" "JDA_LDS_WIN_MAX=112" "JDA_LDS_WIN_MAX=140" "JDA_LDS_WIN_MAX=180" "JDA_LDS_WIN_MAX=220" \
   "JDA_LDS_WIN_MAX=140 JDA_SCAN_P=2" "JDA_SCAN_P=2" "JDA_SCAN_P=0" > $O/variants.txt 2>&1
JDA_LANES=1 JDA_SIDE_STREAM=0 timeout 300 python tools/variants.py "" "JDA_LDS_WIN_MAX=112" "JDA_LDS_WIN_MAX=140" "JDA_LDS_WIN_MAX=180" >> $O/variants.txt 2>&1
export VAR_STEPS=4
timeout 600 python tools/config2_variants.py "" "JDA_LDS_WIN_MAX=130" "JDA_LDS_WIN_MAX=200" "JDA_LDS_WIN_MAX=300" "JDA_LDS_WIN_MAX=420" "JDA_LDS_WIN_MAX=130 JDA_SCAN_P=2" > $O/config2_variants.txt 2>&1
timeout 300 python tools/latency.py > $O/latency.txt 2>&1
cat $O/variants.txt $O/config2_variants.txt $O/latency.txt
