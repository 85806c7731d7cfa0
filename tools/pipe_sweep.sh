#!/bin/bash
# the headline leg alone (tools/pipe.py, PIPE_AHEAD batches queued ahead) under the given environments, twice
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; TAG=${TAG:-pipe_sweep}
for rep in 1 2; do for v in "$@"; do echo "$v: $(env $v timeout 200 python tools/pipe.py 2>&1 | tail -1)"; done; done > gpurun_out/$TAG.log 2>&1
cat gpurun_out/$TAG.log
