#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for i in 1 2; do
for v in 0 140 200 300 500; do echo "== JDA_EXP_C=$v"; JDA_EXP_C=$v python tools/cpp_job.py 5 2>&1 | grep "CPP ragged" | cut -c1-160; done
done
JDA_EXP_C=200 timeout 300 python -m pytest tests/test_cpp_entries.py -x -q -m gpu 2>&1 | tail -2
