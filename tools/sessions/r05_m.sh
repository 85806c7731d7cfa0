#!/bin/bash
# Round 5: k_scan_p for the single-level launches of ragged chunks: parity, then the FDDB-shaped job with and without
TAG=${1:-r05_m}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ragged.py tests/test_fddb.py tests/test_device_post.py tests/test_scan_persistent.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
JDA_SCAN_P=2 timeout 600 python -m pytest tests/test_ragged.py tests/test_fddb.py tests/test_device_post.py -m gpu -x -q >> $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/fddb_job.py 5 "JDA_SCAN_P_RAGGED=0" "" "JDA_SCAN_P_RAGGED=0" "" "JDA_SCAN_P=2" "JDA_RAGGED_TILE_GROW_PCT=100" "JDA_RAGGED_TILE_GROW_PCT=100 JDA_SCAN_P=2" > $O/job.txt 2>&1; cat $O/job.txt
