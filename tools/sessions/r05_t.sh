#!/bin/bash
# r05_t: libjda_dist.so with 2 / 5 / 8 ranks on the one GPU through the RCCL stand-in
mkdir -p gpurun_out/r05_t
timeout 900 python -m pytest tests/test_dist_stub.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r05_t/stub.txt
cat gpurun_out/r05_t/stub.txt
