#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s14; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ragged.py tests/test_cpp_entries.py tests/test_fddb.py tests/test_abi.py -x -q -m gpu > $O/tests.txt 2>&1; tail -8 $O/tests.txt
for v in 0 200; do JDA_EXP_C=$v python tools/cpp_job.py 5 2>&1 | grep "CPP ragged\|C call" | cut -c1-170; done
for v in 0 200; do JDA_EXP_C=$v python tools/cpp_job.py 5 2>&1 | grep "CPP ragged\|C call" | cut -c1-170; done
