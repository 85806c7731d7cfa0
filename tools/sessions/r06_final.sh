#!/bin/bash
# what the driver runs at round end, on the final tree: smoke(), the GPU suite, bench.py with its arguments;
# then the suite on the bounds-check build of the device code, and the two soaks
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_final; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/suite.txt 2>&1; grep -E "passed|failed|error" $O/suite.txt | tail -3
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err ) 2>&1 | grep real
tail -1 $O/bench_driver_args.json | cut -c1-400
JDA_LIB_PATH=jda_amd/libjda_bounds.so timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/suite_bounds.txt 2>&1; grep -E "passed|failed|error" $O/suite_bounds.txt | tail -3
JDA_LIB_PATH=jda_amd/libjda_bounds.so python tools/bounds_selftest.py > $O/bounds_selftest.txt 2>&1; JDA_LIB_PATH=jda_amd/libjda_bounds.so JDA_BOUNDS_TEST_SHRINK=20000 python tools/bounds_selftest.py >> $O/bounds_selftest.txt 2>&1; tail -3 $O/bounds_selftest.txt
timeout 200 python tools/stress.py 40 > $O/stress.txt 2>&1; tail -2 $O/stress.txt
timeout 200 python tools/stress_mixed.py 40 > $O/stress_mixed.txt 2>&1; tail -2 $O/stress_mixed.txt
