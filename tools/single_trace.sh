#!/bin/bash
# kernel trace of single-frame jdaDetect calls (640x480, plan cached): where the call's GPU time goes
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-single}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
cat > /tmp/single.py <<PY
import os, sys
sys.path.insert(0, "$R")
from jda_amd import synth, api
calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
mp = os.path.join(synth.cache_dir(), "model_5_540_27_4_cascade_s1.model")
if not os.path.exists(mp):
    m = synth.make_model(5, 540, 27, 4, seed=1); synth.calibrate_thresholds(m, calib); m.save(mp, 8)
c = api.Cascador(mp)
f = synth.make_frames(4, 640, 480, seed=1)
for i in range(60): c.detect(f[i % 4])
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python /tmp/single.py > $O/run.txt 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) k_ > $O/kernel_trace.txt
find $O -name "*.db" -delete; rm -rf $O/kt
head -12 $O/kernel_trace.txt | cut -c1-200
grep "us grid" $O/kernel_trace.txt | tail -12 | cut -c1-150
