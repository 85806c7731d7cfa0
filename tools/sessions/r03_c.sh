#!/bin/bash
# round 3, session c: full GPU suite after the lane-pool refactor, ragged chunk-size sweep, fddb loop with one cascador
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03c_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r03c_pytest.txt
tail -15 gpurun_out/r03c_pytest.txt
for cw in 3000000 6000000 12000000; do
  echo "ragged_chunk_windows $cw"; JDA_RAGGED_CHUNK_WINDOWS=$cw timeout 300 python tools/ragged_bench.py --variants device,packed --reps 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-8s %.0f images/s %.2f ms  %.3e windows/s' % (d['variant'], d['value'], d['ms_per_job'], d['windows_per_s']))"
done | tee gpurun_out/r03c_ragged_sweep.txt
timeout 600 python tools/fddb_bench.py --dialect c --threads 8 > gpurun_out/r03c_fddb_loop.json 2> gpurun_out/r03c_fddb_loop.err; tail -2 gpurun_out/r03c_fddb_loop.json
timeout 300 python tools/latency.py > gpurun_out/r03c_latency.txt 2>&1; cat gpurun_out/r03c_latency.txt
