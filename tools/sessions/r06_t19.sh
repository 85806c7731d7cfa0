#!/bin/bash
# r06_t19: persistent scan's workgroup size and buckets under the final schedule (two tickets, no side stream next to another pass)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t19; mkdir -p $O; cd $R
run() { local label="$1"; shift; echo -n "$label: "; env "$@" PIPE_STEPS=120 PIPE_AHEAD=1 python tools/pipe.py 2>>$O/log.txt | tail -1; }
for i in 1 2; do
run "block 768 (product)" X=1
run "block 640" JDA_SCAN_P_BLOCK=640
run "block 896" JDA_SCAN_P_BLOCK=896
run "block 1024" JDA_SCAN_P_BLOCK=1024
run "handoff 96" JDA_HANDOFF=96
run "handoff 160" JDA_HANDOFF=160
run "scan_p off" JDA_SCAN_P=0
run "lanes_reverse irrelevant; merge_blocks 0" JDA_MERGE_BLOCKS=0
done
