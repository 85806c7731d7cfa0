#!/bin/bash
# the whole GPU suite on the product build, then on the bounds-check build of the device code
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_suite; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/suite.txt 2>&1; grep -E "passed|failed|error" $O/suite.txt | tail -3
JDA_LIB_PATH=jda_amd/libjda_bounds.so timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/suite_bounds.txt 2>&1; grep -E "passed|failed|error" $O/suite_bounds.txt | tail -3
JDA_LIB_PATH=jda_amd/libjda_bounds.so python tools/bounds_selftest.py > $O/bounds_selftest.txt 2>&1; JDA_LIB_PATH=jda_amd/libjda_bounds.so JDA_BOUNDS_TEST_SHRINK=20000 python tools/bounds_selftest.py >> $O/bounds_selftest.txt 2>&1; tail -3 $O/bounds_selftest.txt
