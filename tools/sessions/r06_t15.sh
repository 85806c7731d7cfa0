#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t15; mkdir -p $O; cd $R
for d in 3 2; do python tools/experiments/r06_step_transient.py $d 20 2>>$O/log.txt; done
python tools/experiments/r06_step_transient.py 2 100 2>>$O/log.txt | cut -c1-60
