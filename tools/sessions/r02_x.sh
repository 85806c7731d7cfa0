#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02_x; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd $R; python tools/x_allpass.py --frames 1 --steps 2 > $O/x_allpass.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -- python $R/tools/x_allpass.py --frames 1 --steps 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -- python $R/tools/x_allpass.py --frames 1 --steps 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/x_allpass.py --frames 1 --steps 1 > /dev/null 2>&1
cd $R
python tools/rocpd_pmc.py $(find $O/f -name "*.db" | head -1) k_finish > $O/x_pmc_hbm.txt; python tools/rocpd_pmc.py $(find $O/w -name "*.db" | head -1) k_finish >> $O/x_pmc_hbm.txt
python tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) > $O/x_kernel_trace.txt
rm -rf $O/f $O/w $O/kt
grep -v amdgpu $O/x_allpass.txt; cat $O/x_pmc_hbm.txt | cut -c1-150; head -6 $O/x_kernel_trace.txt | cut -c1-150
