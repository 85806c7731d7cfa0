#!/bin/bash
# r06_t13: sub-batch lanes of ONE synchronous call on placed streams (configs[1] 256 x 640x480, configs[2] 256 x 1080p)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t13; mkdir -p $O; cd $R
VAR_STEPS=30 python tools/variants.py "" "JDA_LANES=1" "JDA_LANES=3" "JDA_LANES=4" "JDA_LANES=2 JDA_SCAN_P_SLOTS=6" "JDA_LANES=3 JDA_SCAN_P_SLOTS=6" "" 2>>$O/log.txt | cut -c1-110
for l in 2 1 3 4 2; do echo -n "configs[2] lanes $l: "; JDA_LANES=$l python tools/config2.py 2>>$O/log.txt | tail -1 | cut -c150-330; done
