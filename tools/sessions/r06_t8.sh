#!/bin/bash
# r06_t8: settings of the headline pipeline that were tuned under an accidental stream -> queue deal, re-measured on placed streams
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t8; mkdir -p $O; cd $R
run() { local label="$1"; shift; echo -n "$label: "; env "$@" PIPE_STEPS=120 python tools/pipe.py 2>>$O/log.txt | tail -1; }
for i in 1 2; do
run "ahead2 slots5 (product)" PIPE_AHEAD=2
run "ahead1 slots5" PIPE_AHEAD=1
run "ahead3 slots5" PIPE_AHEAD=3
run "ahead2 slots6" PIPE_AHEAD=2 JDA_SCAN_P_SLOTS=6
run "ahead2 slots4" PIPE_AHEAD=2 JDA_SCAN_P_SLOTS=4
run "ahead1 slots6" PIPE_AHEAD=1 JDA_SCAN_P_SLOTS=6
run "ahead2 no side stream" PIPE_AHEAD=2 JDA_SIDE_STREAM=0
run "ahead2 hwq8" PIPE_AHEAD=2 GPU_MAX_HW_QUEUES=8
run "ahead2 dynamic queues" PIPE_AHEAD=2 DEBUG_HIP_DYNAMIC_QUEUES=1
done
