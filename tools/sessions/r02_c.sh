#!/bin/bash
O=gpurun_out/r02_c; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not config4" 2>&1 | tail -5 > $O/tests.txt
S="JDA_LANES=1 JDA_SIDE_STREAM=0"
python tools/variants.py "" "$S" "$S JDA_FIRST_PHASE=16" "$S JDA_ILP8_LDS=40000" "$S JDA_ILP8_LDS=20000" \
  "$S JDA_LDS_WIN_MAX=100" "$S JDA_LDS_WIN_MAX=140" "$S JDA_LDS_WIN_MAX=1000 JDA_ILP8_WIDE=1" \
  "$S JDA_TILES=57:16x16,71:16x16,88:16x16" "$S JDA_TILES=57:30x17,71:21x12,88:14x17" "$S JDA_TILES=46:32x16,57:32x15" \
  "$S JDA_TILES=71:16x16,88:16x16 JDA_ILP8_LDS=30000" "$S JDA_CP_MAX=256" "$S JDA_CP_MAX=64" > $O/variants.txt 2>&1
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
JDA_LANES=1 JDA_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $R/$O/kt -- python $R/tools/variants.py "" "JDA_ILP8_LDS=20000" > /dev/null 2>&1
cd $R; f=$(find $O/kt -name "*.db" | head -1); python tools/rocpd_summary.py $f k_ > $O/kt_stats.txt; rm -rf $O/kt
cat $O/tests.txt $O/variants.txt; head -50 $O/kt_stats.txt | cut -c1-160
