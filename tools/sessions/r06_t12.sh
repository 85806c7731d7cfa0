#!/bin/bash
# r06_t12: a rank's shard (1/8 of the FDDB-sized job) as one chunk or cut into several, on placed streams
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t12; mkdir -p $O; cd $R
run() { local label="$1"; shift; echo -n "$label: "; env "$@" python tools/shard_job.py 15 2>>$O/log.txt | tail -1 | cut -c1-130; }
for i in 1 2; do
run "one chunk, side stream (product)" X=1
run "three chunks" JDA_RAGGED_SINGLE_WINDOWS=0
run "two chunks" JDA_RAGGED_SINGLE_WINDOWS=0 JDA_RAGGED_SPLIT=2
run "four chunks" JDA_RAGGED_SINGLE_WINDOWS=0 JDA_RAGGED_SPLIT=4 JDA_RAGGED_CHUNK_MIN_WINDOWS=1000000
run "one chunk, no side" JDA_SIDE_STREAM=0
done
