#!/bin/bash
# BASELINE.json configs[4] all-pass regime: FETCH_SIZE of the weight-row gather calibrated on the same access pattern
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03_x; mkdir -p $O
cd $R; python tools/pmc_calib.py gather > /dev/null 2>&1     # builds libpmc_calib.so
timeout 300 python tools/x_allpass.py --frames 1 --steps 2 2>&1 | grep -v amdgpu > $O/x_allpass.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/cf -- python $R/tools/pmc_calib.py gather > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum -d $O/cr -- python $R/tools/pmc_calib.py gather > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/xf -- python $R/tools/x_allpass.py --frames 1 --steps 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum -d $O/xr -- python $R/tools/x_allpass.py --frames 1 --steps 1 > /dev/null 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
( cat $O/x_allpass.txt; python tools/x_traffic.py $(db cf) $(db xf) $O/x_allpass.txt $(db cr) $(db xr) ) > $O/r03_x_allpass.txt 2>&1
rm -rf $O/cf $O/cr $O/xf $O/xr
cat $O/r03_x_allpass.txt
