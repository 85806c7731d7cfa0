#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for i in 1 2 3; do for lib in jda_amd/libjda_a.so jda_amd/libjda.so; do
  echo -n "$lib: "; JDA_LIB_PATH=$lib python tools/cpp_job.py 5 2>&1 | grep "CPP ragged" | cut -c38-130
  echo -n "$lib: "; JDA_LIB_PATH=$lib python tools/cpp_bench.py 256 2>&1 | grep "uniform 256 x 640x480, resident" | cut -c40-140
done; done
JDA_LIB_PATH=jda_amd/libjda.so timeout 600 python -m pytest tests/test_cpp_entries.py tests/test_dialect_differential.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
