#!/bin/bash
# r05_y: the randomised call-surface test alone, product build
mkdir -p gpurun_out/r05_y
timeout 1200 python -m pytest tests/test_fuzz_calls.py -x -q -m gpu 2>&1 | tail -40 | tee gpurun_out/r05_y/fuzz.txt
