#!/bin/bash
# r06_t20: hand-off (carts of stage 0 the scan evaluates) under the final schedule; headline pipeline, sync call, FDDB-sized job, shard, configs[2]
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t20; mkdir -p $O; cd $R
for i in 1 2 3; do for h in 128 144 160 176 192 224 256; do
  echo -n "handoff $h pipe2: "; JDA_HANDOFF=$h PIPE_STEPS=120 PIPE_AHEAD=1 python tools/pipe.py 2>>$O/log.txt | tail -1
done; done
for h in 128 160 192; do
  echo -n "handoff $h pipe3: "; JDA_HANDOFF=$h PIPE_STEPS=120 PIPE_AHEAD=2 python tools/pipe.py 2>>$O/log.txt | tail -1
  JDA_HANDOFF=$h VAR_STEPS=30 python tools/variants.py "" "JDA_LANES=1 JDA_SIDE_STREAM=0" 2>>$O/log.txt | cut -c1-120
  echo -n "handoff $h C job: "; JDA_HANDOFF=$h python tools/fddb_job.py 5 2>>$O/log.txt | tail -1 | cut -c60-130
  echo -n "handoff $h shard: "; JDA_HANDOFF=$h python tools/shard_job.py 15 2>>$O/log.txt | tail -1 | cut -c40-130
  echo -n "handoff $h configs[2]: "; JDA_HANDOFF=$h python tools/config2.py 2>>$O/log.txt | tail -1 | cut -c150-260
  echo -n "handoff $h cpp job: "; JDA_HANDOFF=$h python tools/cpp_job.py 5 2>>$O/log.txt | grep "CPP ragged job" | cut -c38-80
  echo -n "handoff $h latency: "; JDA_HANDOFF=$h python tools/latency.py 2>>$O/log.txt | head -2 | tr '\n' ' ' | cut -c1-150; echo
done
