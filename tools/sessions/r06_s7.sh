#!/bin/bash
# k_finish: 5 / 9 groups of 64 carts walked per round (2 rounds / 1 round per stage instead of 3)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s7; mkdir -p $O; cd $R
run() { echo "== $*"; for r in 0 3; do env "$@" timeout 120 python tools/shard_job.py 20 8 $r 2>&1 | tail -1; done; env "$@" PIPE_STEPS=60 PIPE_AHEAD=2 timeout 120 python tools/pipe.py 2>&1 | tail -1; }
export JDA_EXP_A=1
(run JDA_EXP_B=0; run JDA_EXP_B=5; run JDA_EXP_B=9; run JDA_EXP_B=4; run JDA_EXP_B=0; run JDA_EXP_B=5; run JDA_EXP_B=9) > $O/ab.txt 2>&1
VAR_STEPS=20 timeout 300 python tools/variants.py "JDA_LANES=1 JDA_SIDE_STREAM=0" "JDA_LANES=1 JDA_SIDE_STREAM=0 JDA_EXP_B=5" "JDA_LANES=1 JDA_SIDE_STREAM=0 JDA_EXP_B=9" "" "JDA_EXP_B=5" "JDA_EXP_B=9" >> $O/ab.txt 2>&1
grep -v amdgpu $O/ab.txt
