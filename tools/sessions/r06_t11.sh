#!/bin/bash
# r06_t11: LDS pixel tiles for dialect CPP's bigger windows (its 5-pixel step keeps them dense), re-measured on placed streams
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t11; mkdir -p $O; cd $R
for i in 1 2; do for v in 100 140 200 300; do
  echo -n "lds_win_max(cpp) $v | cpp job: "; JDA_X_LDS_WIN_MAX_CPP=$v python tools/cpp_job.py 5 2>>$O/log.txt | grep "CPP ragged job" | cut -c38-140
  echo -n "lds_win_max(cpp) $v | uniform: "; JDA_X_LDS_WIN_MAX_CPP=$v python tools/cpp_bench.py 256 2>>$O/log.txt | grep "uniform 256 x 640x480, resident" | cut -c40-150
done; done
