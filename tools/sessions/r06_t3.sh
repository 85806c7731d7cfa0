#!/bin/bash
# r06_t3: the runtime's deal of streams to hardware queues (tools/experiments/hwq_probe.hip)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t3; mkdir -p $O; cd $R
P=tools/experiments/hwq_probe.bin
{ echo "## default, 10 streams"; $P 10 1
  echo "## default, 10 streams, null stream not used"; $P 10 0
  echo "## GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 $P 10 1
  echo "## DEBUG_HIP_DYNAMIC_QUEUES=0"; DEBUG_HIP_DYNAMIC_QUEUES=0 $P 10 1
  echo "## DEBUG_HIP_DYNAMIC_QUEUES=1"; DEBUG_HIP_DYNAMIC_QUEUES=1 $P 10 1
  echo "## priorities 0/hi/lo in turn"; $P 9 1 1
} > $O/hwq_probe.txt 2>&1
cat $O/hwq_probe.txt
