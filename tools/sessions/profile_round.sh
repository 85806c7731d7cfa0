#!/bin/bash
# End-of-round measurements on the GPU box: bench line, kernel traces, PMC passes.
#   gpurun --timeout 1500 -- 'tools/sessions/profile_round.sh r01_e'
# Outputs under gpurun_out/<tag>/ ; summaries are made from the rocpd databases by
# tools/rocpd_summary.py / rocpd_pmc.py / make_traffic_json.py (run them where the files are).
TAG=${1:-r01_x}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu --no-allpass"
# kernel trace: one lane (k_scan launches back to back: durations comparable with roofline.kernel_ms_per_step)
JDA_LANES=1 JDA_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/kt_lane1 -- $B --steps 10 --warmup 2 > $O/bench_kt_lane1.json 2> /dev/null
# kernel trace: default (two lanes, kernels of the two sub-batches overlap)
rocprofv3 --kernel-trace --stats -d $O/kt -- $B --steps 10 --warmup 2 > $O/bench_kt.json 2> /dev/null
# HBM traffic: separate passes, one counter each (1 warm-up + 4 steps + roofline leg 1 + 4 = 10 passes of the batch)
JDA_LANES=1 JDA_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -- $B --steps 4 --warmup 1 > /dev/null 2>&1
JDA_LANES=1 JDA_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -- $B --steps 4 --warmup 1 > /dev/null 2>&1
# SQ counters, cascade regime
JDA_LANES=1 JDA_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq1 -- $B --steps 4 --warmup 1 > /dev/null 2>&1
JDA_LANES=1 JDA_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM -d $O/pmc_sq2 -- $B --steps 4 --warmup 1 > /dev/null 2>&1
# SQ counters, all-pass regime (k_stage)
A="python $R/tools/allpass.py --steps 1 --batch 64"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/pmc_stage -- $A > $O/allpass.txt 2>&1
cd $R
for d in kt_lane1 kt; do f=$(find $O/$d -name "*.db" | head -1); python tools/rocpd_summary.py $f k_ > $O/${d}_stats.txt; done
for d in pmc_f pmc_w pmc_sq1 pmc_sq2; do f=$(find $O/$d -name "*.db" | head -1); python tools/rocpd_pmc.py $f > $O/${d}.txt; done
f=$(find $O/pmc_stage -name "*.db" | head -1); python tools/rocpd_pmc.py $f k_stage > $O/pmc_stage.txt
python tools/allpass.py --steps 2 --batch 64 > $O/allpass_64.txt 2>&1
python tools/allpass.py --steps 2 --batch 64 JDA_DENSE=0 >> $O/allpass_64.txt 2>&1
python tools/allpass.py --steps 2 --batch 64 --dims 5,540,5,4 >> $O/allpass_64.txt 2>&1
python tools/allpass.py --steps 2 --batch 64 --dims 5,540,5,4 JDA_DENSE=0 >> $O/allpass_64.txt 2>&1
python tools/latency.py > $O/latency.txt 2>&1
# keep only the databases that the traffic json needs (<= 64 MiB comes back)
find $O -name "*.db" ! -path "*pmc_f*" ! -path "*pmc_w*" -delete
du -sh $O
