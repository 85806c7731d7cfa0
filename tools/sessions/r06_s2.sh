#!/bin/bash
# fp64 scan: workgroup stamps of one chunk (timing build) + SQ counters of the CPP job's launches on one lane
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s2; mkdir -p $O; cd $R
JDA_LIB_PATH=jda_amd/libjda_timing.so timeout 200 python tools/scan_timing_cpp.py 380 > $O/scan_timing_cpp.txt 2>&1
JDA_LIB_PATH=jda_amd/libjda_timing.so timeout 200 python tools/scan_timing_ragged.py > $O/scan_timing_ragged.txt 2>&1
cd /tmp; export TMPDIR=/tmp
JDA_RAGGED_LANES=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/sq1 -- python $R/tools/cpp_job.py 2 > /dev/null 2>&1
JDA_RAGGED_LANES=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM -d $O/sq2 -- python $R/tools/cpp_job.py 2 > /dev/null 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocpd_pmc.py $(db sq1) > $O/pmc_sq.txt; python tools/rocpd_pmc.py $(db sq2) >> $O/pmc_sq.txt
rm -rf $O/sq1 $O/sq2
grep -v amdgpu $O/scan_timing_cpp.txt | tail -30
