#!/bin/bash
# all-pass regime with the shipped dimensions (k_stage, dense mode): kernel trace, SQ and FETCH_SIZE / WRITE_SIZE passes
#   gpurun -- 'bash tools/sessions/r04_allpass.sh'  -> gpurun_out/r04_allpass/*
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_allpass; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
A="python $R/tools/allpass.py --batch 64 --steps 2"
timeout 300 $A 2>/dev/null | grep gpu_ms > $O/rates.txt; cat $O/rates.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- $A > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -- $A > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -- $A > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/s1 -- $A > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $O/s2 -- $A > /dev/null 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocpd_summary.py $(db kt) | head -12 | cut -c1-200 > $O/kernel_trace_stats.txt
{ python tools/rocpd_pmc.py $(db pf); python tools/rocpd_pmc.py $(db pw); python tools/rocpd_pmc.py $(db s1); python tools/rocpd_pmc.py $(db s2); } | grep "k_stage" | cut -c1-190 > $O/pmc_k_stage.txt
find $O -name "*.db" -delete; rm -rf $O/kt $O/pf $O/pw $O/s1 $O/s2
cat $O/kernel_trace_stats.txt; cat $O/pmc_k_stage.txt
