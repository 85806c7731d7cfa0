#!/bin/bash
# r06_t18: longer soaks and randomised parity runs on the final tree
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t18; mkdir -p $O; cd $R
timeout 400 python tools/stress_mixed.py 150 2>&1 | tail -1
timeout 400 python tools/stress.py 100 2>&1 | tail -1
for s in 21 22; do timeout 300 python tools/fuzz_c_product.py $s 110 2>&1 | tail -1; done
for s in 23 24; do timeout 300 python tools/fuzz_cpp_product.py $s 110 2>&1 | tail -1; done
python tools/fuzz_model.py 2>&1 | tail -2
