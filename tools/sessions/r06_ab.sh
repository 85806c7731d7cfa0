#!/bin/bash
# A/B on one box: jda_amd/libjda_base.so (the library as of commit 36b267c, start of this session) vs the working tree's
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for i in 1 2 3; do
  for lib in jda_amd/libjda_base.so jda_amd/libjda.so; do
    echo "== $lib"
    JDA_LIB_PATH=$lib python tools/cpp_job.py 5 2>&1 | grep "CPP ragged" | cut -c1-150
    JDA_LIB_PATH=$lib python tools/fddb_job.py 10 "" "" 2>&1 | grep "per job" | tail -1 | cut -c50-170
  done
done
