#!/bin/bash
# r06: the weight-row gather regime re-measured on this round's device code, twice: BASELINE configs[4] (T=7: W = 243.7 MB,
# inside the 256 MB Infinity Cache) and the same model with T=14 (W = 487.4 MB, beyond it; a 960x540 frame keeps it short).
# FETCH_SIZE calibrated on the same access pattern (tools/pmc_calib.py gather); JSONs stamped with the device-code hash.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_x; mkdir -p $O
cd $R; python tools/pmc_calib.py gather > /dev/null 2>&1     # builds libpmc_calib.so
timeout 300 python tools/x_allpass.py --frames 1 --steps 2 2>&1 | grep -v amdgpu > $O/x7.txt
timeout 300 python tools/x_allpass.py --stages 14 --size 960x540 --frames 1 --steps 2 2>&1 | grep -v amdgpu > $O/x14.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/cf -- python $R/tools/pmc_calib.py gather > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum -d $O/cr -- python $R/tools/pmc_calib.py gather > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/xf7 -- python $R/tools/x_allpass.py --frames 1 --steps 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum -d $O/xr7 -- python $R/tools/x_allpass.py --frames 1 --steps 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/xf14 -- python $R/tools/x_allpass.py --stages 14 --size 960x540 --frames 1 --steps 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum -d $O/xr14 -- python $R/tools/x_allpass.py --stages 14 --size 960x540 --frames 1 --steps 1 > /dev/null 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
( cat $O/x7.txt; JDA_X_SOURCE=tools/sessions/r06_x.sh JDA_X_JSON=x_allpass_traffic.json python tools/x_traffic.py $(db cf) $(db xf7) $O/x7.txt $(db cr) $(db xr7) ) > $O/r06_x_allpass.txt 2>&1
( cat $O/x14.txt; JDA_X_SOURCE=tools/sessions/r06_x.sh JDA_X_JSON=x_allpass_T14_traffic.json python tools/x_traffic.py $(db cf) $(db xf14) $O/x14.txt $(db cr) $(db xr14) ) > $O/r06_x_allpass_T14.txt 2>&1
rm -rf $O/cf $O/cr $O/xf7 $O/xr7 $O/xf14 $O/xr14
cat $O/r06_x_allpass.txt $O/r06_x_allpass_T14.txt
