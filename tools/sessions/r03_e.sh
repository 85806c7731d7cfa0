#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03e_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r03e_pytest.txt
tail -4 gpurun_out/r03e_pytest.txt
for v in "" "JDA_SIDE_SMALL=0"; do echo "== $v"; env $v timeout 300 python tools/latency.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03e_latency.txt
bash tools/single_trace.sh single_e
