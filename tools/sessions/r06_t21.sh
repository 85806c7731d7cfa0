#!/bin/bash
# r06_t21: a later hand-off for the persistent scan's levels only (scan_p_handoff), under the final schedule
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t21; mkdir -p $O; cd $R
for i in 1 2; do for h in 0 144 160 192 256; do
  echo -n "scan_p_handoff $h pipe2: "; JDA_SCAN_P_HANDOFF=$h PIPE_STEPS=120 PIPE_AHEAD=1 python tools/pipe.py 2>>$O/log.txt | tail -1
done; done
for h in 0 160; do
  echo -n "scan_p_handoff $h pipe3: "; JDA_SCAN_P_HANDOFF=$h PIPE_STEPS=120 PIPE_AHEAD=2 python tools/pipe.py 2>>$O/log.txt | tail -1
  JDA_SCAN_P_HANDOFF=$h VAR_STEPS=30 python tools/variants.py "" "JDA_LANES=1 JDA_SIDE_STREAM=0" 2>>$O/log.txt | cut -c1-120
  echo -n "scan_p_handoff $h C job: "; JDA_SCAN_P_HANDOFF=$h python tools/fddb_job.py 5 2>>$O/log.txt | tail -1 | cut -c60-130
  echo -n "scan_p_handoff $h shard: "; JDA_SCAN_P_HANDOFF=$h python tools/shard_job.py 15 2>>$O/log.txt | tail -1 | cut -c40-130
done
