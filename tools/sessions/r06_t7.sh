#!/bin/bash
# r06_t7: chunks of a ragged job in flight (lanes) and chunk sizes, re-measured with every lane on a hardware queue of its own
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t7; mkdir -p $O; cd $R
run() { local label="$1"; shift
  echo -n "$label | cpp job: "; env "$@" python tools/cpp_job.py 5 2>>$O/log.txt | grep "CPP ragged job" | cut -c38-80
  echo -n "$label | C job: "; env "$@" python tools/fddb_job.py 5 2>>$O/log.txt | tail -1 | cut -c60-130; }
for l in 3 2 4 5 3; do run "lanes$l" JDA_X_RAGGED_LANES=$l; done
run "lanes4 chunks 3M/6M" JDA_X_RAGGED_LANES=4 JDA_RAGGED_CHUNK_WINDOWS=3000000 JDA_RAGGED_CHUNK_WINDOWS_CPP=6000000
run "lanes4 chunks 6M/12M" JDA_X_RAGGED_LANES=4 JDA_RAGGED_CHUNK_WINDOWS=6000000 JDA_RAGGED_CHUNK_WINDOWS_CPP=12000000
run "lanes3 chunks 6M/12M" JDA_X_RAGGED_LANES=3 JDA_RAGGED_CHUNK_WINDOWS=6000000 JDA_RAGGED_CHUNK_WINDOWS_CPP=12000000
run "lanes2 chunks 6M/12M" JDA_X_RAGGED_LANES=2 JDA_RAGGED_CHUNK_WINDOWS=6000000 JDA_RAGGED_CHUNK_WINDOWS_CPP=12000000
run "lanes4 hwq8" JDA_X_RAGGED_LANES=4 GPU_MAX_HW_QUEUES=8
