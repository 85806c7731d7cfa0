#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for i in 1 2 3; do for g in 0 1 2; do echo -n "TILE_GRANULE=$g "; JDA_TILE_GRANULE=$g PIPE_STEPS=80 PIPE_AHEAD=2 python tools/pipe.py 2>&1 | tail -1; done; done
VAR_STEPS=20 timeout 300 python tools/variants.py "JDA_LANES=1 JDA_SIDE_STREAM=0" "JDA_LANES=1 JDA_SIDE_STREAM=0 JDA_TILE_GRANULE=1" "JDA_LANES=1 JDA_SIDE_STREAM=0 JDA_TILE_GRANULE=2" "JDA_LANES=1 JDA_SIDE_STREAM=0 JDA_TILE_GRANULE=1 JDA_DEBUG_TILES=1" 2>&1 | grep -v amdgpu
for g in 0 1 2; do echo "== config2 TILE_GRANULE=$g"; JDA_TILE_GRANULE=$g timeout 200 python tools/config2_variants.py "" 2>&1 | tail -1; done
for g in 0 1 2; do echo -n "fddb job TILE_GRANULE=$g: "; JDA_TILE_GRANULE=$g python tools/fddb_job.py 10 "" "" 2>&1 | grep "per job" | tail -1 | cut -c50-130; done
