#!/bin/bash
# r06_t17: k_finish's windows of a CU started a fraction of a stage apart (JDA_X_FIN_STAGGER = shader clocks per wave slot)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t17; mkdir -p $O; cd $R
for s in 0 1500 3000 5000 8000 0; do
  echo "== stagger $s"
  echo -n "  shard: "; JDA_X_FIN_STAGGER=$s python tools/shard_job.py 15 2>>$O/log.txt | tail -1 | cut -c40-130
  JDA_X_FIN_STAGGER=$s VAR_STEPS=20 python tools/variants.py "JDA_LANES=1 JDA_SIDE_STREAM=0" "" 2>>$O/log.txt | cut -c1-100
  echo -n "  pipe: "; JDA_X_FIN_STAGGER=$s PIPE_STEPS=120 PIPE_AHEAD=1 python tools/pipe.py 2>>$O/log.txt | tail -1
done
cd /tmp; export TMPDIR=/tmp
for s in 0 3000; do
  JDA_X_FIN_STAGGER=$s JDA_LANES=1 JDA_SIDE_STREAM=0 VAR_STEPS=10 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$s -- python $R/tools/variants.py "" > /dev/null 2>&1
  echo "== stagger $s, launches back to back"; python $R/tools/rocpd_summary.py $(find $O/kt_$s -name "*.db" | head -1) k_fin | head -4 | cut -c1-150
  rm -rf $O/kt_$s
done
