#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ragged.py -x -q -s > gpurun_out/r03b_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r03b_pytest.txt
tail -25 gpurun_out/r03b_pytest.txt
timeout 600 python tools/ragged_bench.py > gpurun_out/r03b_ragged.jsonl 2> gpurun_out/r03b_ragged.err; echo "ragged rc $?"
cat gpurun_out/r03b_ragged.jsonl; tail -5 gpurun_out/r03b_ragged.err
