#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for i in 1 2; do for lib in jda_amd/libjda_base.so jda_amd/libjda.so; do
  echo -n "$lib: "; JDA_LIB_PATH=$lib python tools/cpp_bench.py 256 2>&1 | grep "uniform 256 x 640x480, resident" | cut -c40-150
  echo -n "$lib: "; JDA_LIB_PATH=$lib PIPE_STEPS=80 PIPE_AHEAD=2 python tools/pipe.py 2>&1 | tail -1
done; done
VAR_STEPS=20 python tools/variants.py "JDA_LANES=1 JDA_SIDE_STREAM=0" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cpp_entries.py tests/test_scan_persistent.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
