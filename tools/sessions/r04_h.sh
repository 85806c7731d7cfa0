#!/bin/bash
# configs[4] all-pass under the given environments (one process each), then its fabric request counters for the first two
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out; TAG=${TAG:-r04_h}
for v in "$@"; do echo "=== $v"; env $v timeout 300 python tools/x_allpass.py --frames 1 --steps 2 2>&1 | grep -v amdgpu.ids; done > gpurun_out/$TAG.log 2>&1
cd /tmp; export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1)); [ $i -gt 2 ] && break
  rm -rf /tmp/xr
  env $v timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d /tmp/xr -- python $R/tools/x_allpass.py --frames 1 --steps 1 > /dev/null 2>&1
  echo "=== counters: $v" >> $R/gpurun_out/$TAG.log
  python $R/tools/rocpd_pmc.py $(find /tmp/xr -name "*.db" | head -1) k_finish 2>&1 | cut -c1-150 >> $R/gpurun_out/$TAG.log
done
cat $R/gpurun_out/$TAG.log
