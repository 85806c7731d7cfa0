#!/bin/bash
# BASELINE configs[2] at its stated size: rates (sync / submit-wait), kernel trace, SQ and FETCH_SIZE / WRITE_SIZE passes
#   gpurun -- 'bash tools/sessions/r04_config2.sh'   -> gpurun_out/r04_config2/*
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_config2; mkdir -p $O
cd $R
{ timeout 300 python tools/config2.py sync 5; timeout 300 python tools/config2.py pipe 5;
  JDA_LANES=1 JDA_SIDE_STREAM=0 timeout 300 python tools/config2.py sync 5; } 2>/dev/null | grep "^{" > $O/rates.jsonl
cat $O/rates.jsonl
cd /tmp; export TMPDIR=/tmp
S="JDA_LANES=1 JDA_SIDE_STREAM=0"
env $S timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/config2.py sync 3 > /dev/null 2>&1
env $S timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -- python $R/tools/config2.py sync 3 > /dev/null 2>&1
env $S timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -- python $R/tools/config2.py sync 3 > /dev/null 2>&1
env $S timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/s1 -- python $R/tools/config2.py sync 3 > /dev/null 2>&1
env $S timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU TA_BUSY_avr TA_TA_BUSY_sum -d $O/s2 -- python $R/tools/config2.py sync 3 > /dev/null 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocpd_summary.py $(db kt) k_ | cut -c1-200 > $O/kernel_trace_stats.txt
{ python tools/rocpd_pmc.py $(db pf); python tools/rocpd_pmc.py $(db pw); } | grep "k_" | cut -c1-190 > $O/pmc_hbm.txt
{ python tools/rocpd_pmc.py $(db s1); python tools/rocpd_pmc.py $(db s2); } | grep "k_" | cut -c1-190 > $O/pmc_sq.txt
find $O -name "*.db" -delete; rm -rf $O/kt $O/pf $O/pw $O/s1 $O/s2
head -40 $O/kernel_trace_stats.txt; cat $O/pmc_hbm.txt
