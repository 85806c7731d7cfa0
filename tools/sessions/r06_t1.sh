#!/bin/bash
# r06_t1: host-side timeline of the dialect-CPP FDDB-shaped job without a profiler (JDA_DEBUG_TIMES=1), and bench.py's wall time
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t1; mkdir -p $O; cd $R
python tools/cpp_job.py 5 > $O/cpp_job_plain.txt 2>&1; grep "CPP ragged job\|C call" $O/cpp_job_plain.txt
JDA_DEBUG_TIMES=1 python tools/cpp_job.py 3 > $O/cpp_job_dbg.txt 2>&1; tail -80 $O/cpp_job_dbg.txt | cut -c1-200
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
