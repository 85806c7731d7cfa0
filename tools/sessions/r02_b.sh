#!/bin/bash
O=gpurun_out/r02_b; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/tests.txt
S="JDA_LANES=1 JDA_SIDE_STREAM=0"
python tools/variants.py "" "$S" "$S JDA_HANDOFF=128" "$S JDA_CP_MAX=256" "$S JDA_LDS_WIN_MAX=100" \
  "$S JDA_LDS_WIN_MAX=100 JDA_HANDOFF=128 JDA_TILES=46:32x16,57:16x16,71:16x16,88:16x16" \
  "$S JDA_TILES=46:32x16,57:30x17,71:28x15,88:24x17" "$S JDA_TILES=46:30x16,57:20x12,71:21x12,88:14x17" \
  "$S JDA_TILES=110:10x9,137:8x7,171:6x5,213:4x4" "$S JDA_TILES=110:17x12,137:13x9,171:7x7" > $O/variants.txt 2>&1
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
JDA_LANES=1 JDA_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $R/$O/kt -- python $R/tools/variants.py "" > /dev/null 2>&1
cd $R; f=$(find $O/kt -name "*.db" | head -1); python tools/rocpd_summary.py $f k_ > $O/kt_stats.txt; rm -rf $O/kt
JDA_LIB_PATH=jda_amd/libjda_timing.so python tools/scan_timing.py > $O/scan_timing.txt 2>&1
cat $O/tests.txt $O/variants.txt; head -40 $O/kt_stats.txt | cut -c1-200; cat $O/scan_timing.txt
