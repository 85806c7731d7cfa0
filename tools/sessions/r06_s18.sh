#!/bin/bash
# k_finish regression with lanes = coordinate pairs / quads: parity, then A/B against jda_amd/libjda_base.so on one box
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s18; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_cpp_entries.py tests/test_ragged.py tests/test_dialect_differential.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
for i in 1 2; do for lib in jda_amd/libjda_base.so jda_amd/libjda.so; do
  echo "== $lib"
  JDA_LIB_PATH=$lib PIPE_STEPS=80 PIPE_AHEAD=2 python tools/pipe.py 2>&1 | tail -1
  JDA_LIB_PATH=$lib VAR_STEPS=20 python tools/variants.py "JDA_LANES=1 JDA_SIDE_STREAM=0" 2>&1 | tail -1 | cut -c60-140
  JDA_LIB_PATH=$lib python tools/cpp_job.py 5 2>&1 | grep "CPP ragged" | cut -c1-120
  JDA_LIB_PATH=$lib python tools/x_allpass.py 2>&1 | grep -v amdgpu | head -2 | cut -c1-200
done; done
