#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s10; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_ragged.py tests/test_cpp_entries.py tests/test_fddb.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python tools/cpp_job.py 5 2>&1 | grep "CPP ragged"
python tools/cpp_job.py 5 2>&1 | grep "CPP ragged"
python tools/fddb_job.py 10 "" "" 2>&1 | grep "per job"
for r in 0 3 7; do timeout 120 python tools/shard_job.py 20 8 $r 2>&1 | tail -1; done
VAR_STEPS=20 timeout 300 python tools/variants.py "JDA_LANES=1 JDA_SIDE_STREAM=0" "JDA_LANES=1 JDA_SIDE_STREAM=0 JDA_TILE_GRANULE=1" "JDA_LANES=1 JDA_SIDE_STREAM=0 JDA_TILE_GRANULE=2" "" "JDA_TILE_GRANULE=1" "JDA_TILE_GRANULE=2" 2>&1 | grep -v amdgpu
for g in 0 1 2; do JDA_TILE_GRANULE=$g PIPE_STEPS=60 PIPE_AHEAD=2 python tools/pipe.py 2>&1 | tail -1; done
for g in 0 1 2; do JDA_TILE_GRANULE=$g python tools/cpp_bench.py 256 2>&1 | grep "resident" | head -2; done
