#!/bin/bash
# r05_v: flake hunt: the GPU suite twice more (second time in reverse file order), then the mixed soak
mkdir -p gpurun_out/r05_v
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r05_v/suite_a.txt
cat gpurun_out/r05_v/suite_a.txt
timeout 1500 python -m pytest $(ls tests/test_*.py | sort -r) -q -m gpu -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r05_v/suite_b.txt
cat gpurun_out/r05_v/suite_b.txt
SOAK_S=60 timeout 300 python tools/stress_mixed.py 2>&1 | tail -6 | tee gpurun_out/r05_v/soak.txt
