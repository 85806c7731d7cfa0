#!/bin/bash
# r06_t2: how the stream -> hardware-queue deal moves the jobs that run several lanes side by side: dialect-CPP FDDB-shaped job,
# dialect-C job, headline pipeline; GPU_MAX_HW_QUEUES, dummy streams created first, lanes at different stream priorities
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t2; mkdir -p $O; cd $R
run() { # label, env...
  local label="$1"; shift
  echo "== $label" >> $O/log.txt
  echo -n "$label | cpp job: "; env "$@" python tools/cpp_job.py 5 2>>$O/log.txt | grep "CPP ragged job" | cut -c38-80
  echo -n "$label | C job: "; env "$@" python tools/fddb_job.py 5 2>>$O/log.txt | tail -1 | cut -c1-120
  echo -n "$label | pipe: "; env "$@" PIPE_STEPS=80 PIPE_AHEAD=2 python tools/pipe.py 2>>$O/log.txt | tail -1
}
run default X=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq6 GPU_MAX_HW_QUEUES=6
run dummy1 JOB_DUMMY=1
run dummy2 JOB_DUMMY=2
run dummy3 JOB_DUMMY=3
run prio1 JDA_X_LANE_PRIO=1
run prio2 JDA_X_LANE_PRIO=2
run prio3 JDA_X_LANE_PRIO=3
run default_again X=1
