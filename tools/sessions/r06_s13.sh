#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s13; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in 3 2 1; do
JDA_RAGGED_LANES=$v timeout 300 rocprofv3 --kernel-trace -d $O/kt -- python $R/tools/cpp_job.py 3 > $O/run.txt 2>&1
echo "== lanes $v: $(grep 'CPP ragged' $O/run.txt | cut -c1-120)"
python $R/tools/job_overlap.py $(find $O/kt -name "*.db" | head -1)
rm -rf $O/kt
done
