#!/bin/bash
# the HBM-traffic passes alone (profiles/hbm_traffic.json carries the hash of the device code they ran on) + one bench line
TAG=${1:-r04_s}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V="python $R/tools/variants.py"; export VAR_STEPS=10; S="JDA_LANES=1 JDA_SIDE_STREAM=0"
env $S timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -- $V "" > /dev/null 2>&1
env $S timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -- $V "" > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/cal_f -- python $R/tools/pmc_calib.py > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/cal_w -- python $R/tools/pmc_calib.py > /dev/null 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocpd_pmc.py $(db pmc_f) > $O/pmc_hbm.txt; python tools/rocpd_pmc.py $(db pmc_w) >> $O/pmc_hbm.txt
python tools/pmc_traffic.py $(db cal_f) $(db cal_w) $(db pmc_f) $(db pmc_w) 13 "$TAG: JDA_LANES=1 JDA_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python tools/variants.py ''" > $O/hbm_traffic.json
cp $O/hbm_traffic.json $R/profiles/hbm_traffic.json      # (so that the bench line below checks its traffic figure against this device code)
find $O -name "*.db" -delete; rm -rf $O/pmc_f $O/pmc_w $O/cal_f $O/cal_w
timeout 300 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-260
