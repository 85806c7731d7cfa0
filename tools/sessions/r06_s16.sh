#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s16; mkdir -p $O; cd $R
VAR_STEPS=20 timeout 300 python tools/variants.py "JDA_LANES=1 JDA_SIDE_STREAM=0" "JDA_LANES=1 JDA_SIDE_STREAM=0 JDA_EXP_A=1" "" "JDA_EXP_A=1" "JDA_LANES=1 JDA_SIDE_STREAM=0" "JDA_LANES=1 JDA_SIDE_STREAM=0 JDA_EXP_A=1" "" "JDA_EXP_A=1" 2>&1 | grep -v amdgpu
for i in 1 2 3; do for g in 0 1; do echo -n "EXP_A=$g "; JDA_EXP_A=$g PIPE_STEPS=80 PIPE_AHEAD=2 python tools/pipe.py 2>&1 | tail -1; done; done
JDA_EXP_A=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_scan_persistent.py -x -q -m gpu 2>&1 | tail -3
