#!/bin/bash
O=gpurun_out/r02_d; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_fddb.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
S="JDA_LANES=1 JDA_SIDE_STREAM=0"
python tools/variants.py "" "JDA_WALK=0" "$S" "$S JDA_WALK=0" "$S JDA_FIN_G2=2" "$S JDA_FIN_G1=2" "$S JDA_FIRST_PHASE=24" "$S JDA_FIRST_PHASE=8" > $O/variants.txt 2>&1
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
JDA_LANES=1 JDA_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $R/$O/kt -- python $R/tools/variants.py "" > /dev/null 2>&1
cd $R; f=$(find $O/kt -name "*.db" | head -1); python tools/rocpd_summary.py $f k_ > $O/kt_stats.txt; rm -rf $O/kt
python tools/latency.py > $O/latency.txt 2>&1
cat $O/tests.txt $O/variants.txt; head -28 $O/kt_stats.txt | cut -c1-180; cat $O/latency.txt
