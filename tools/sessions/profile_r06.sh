#!/bin/bash
# Round-6 measurement set on the GPU box (outputs under gpurun_out/<tag>/, summaries copied to profiles/ by hand):
#   gpurun --timeout 2400 -- 'tools/sessions/profile_r06.sh r06_p'
TAG=${1:-r06_p}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --no-cpu --no-allpass > $O/bench_run2.json 2>> $O/bench.err
cd /tmp; export TMPDIR=/tmp
V="python $R/tools/variants.py"
export VAR_STEPS=10
S="JDA_LANES=1 JDA_SIDE_STREAM=0"
# kernel traces: every launch of a step back to back on one stream / the bench's headline leg (submit/wait, two batches in flight)
env $S timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt1 -- $V "" > $O/kt1_run.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt2 -- python $R/tools/pipe.py > $O/kt2_run.txt 2>/dev/null
# HBM traffic: one counter per pass (13 passes of the batch each: 3 warm-up + 10)
env $S timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -- $V "" > /dev/null 2>&1
env $S timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -- $V "" > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/cal_f -- python $R/tools/pmc_calib.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/cal_w -- python $R/tools/pmc_calib.py > /dev/null 2>&1
# SQ counters
env $S timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/sq1 -- $V "" > /dev/null 2>&1
env $S timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM -d $O/sq2 -- $V "" > /dev/null 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocpd_summary.py $(db kt1) k_ > $O/kernel_trace_one_lane_stats.txt
python tools/rocpd_summary.py $(db kt2) k_ > $O/kernel_trace_stats.txt
python tools/bench_timeline.py $(db kt2) > $O/pipeline_timeline.txt 2>&1
python tools/rocpd_pmc.py $(db pmc_f) > $O/pmc_hbm.txt; python tools/rocpd_pmc.py $(db pmc_w) >> $O/pmc_hbm.txt
python tools/rocpd_pmc.py $(db sq1) > $O/pmc_sq.txt; python tools/rocpd_pmc.py $(db sq2) >> $O/pmc_sq.txt
python tools/pmc_traffic.py $(db cal_f) $(db cal_w) $(db pmc_f) $(db pmc_w) 13 "$TAG: JDA_LANES=1 JDA_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python tools/variants.py ''" > $O/hbm_traffic.json
find $O -name "*.db" -delete; rm -rf $O/kt1 $O/kt2 $O/pmc_f $O/pmc_w $O/cal_f $O/cal_w $O/sq1 $O/sq2
# (configs[4] all-pass regime and the W-beyond-the-Infinity-Cache variant: tools/sessions/r06_x.sh, run on its own)
# dialect CPP: the FDDB-shaped ragged job, kernel trace with its launches alone on one lane and as the product runs it
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktc3 -- python $R/tools/cpp_job.py 5 > $O/cpp_job.txt 2>&1
cd $R
python tools/rocpd_summary.py $(db ktc3) > $O/cpp_job_kernel_trace_stats.txt
python tools/job_overlap.py $(db ktc3) > $O/cpp_job_overlap.txt 2>&1
rm -rf $O/ktc3
timeout 300 python tools/cpp_bench.py 256 > $O/cpp_bench.txt 2>&1
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $O/kts -- python $R/tools/shard_job.py 10 8 0 > /dev/null 2>&1
cd $R
python tools/shard_timeline.py $(db kts) > $O/shard_timeline.txt 2>&1
rm -rf $O/kts
for r in 0 1 2 3 4 5 6 7; do timeout 120 python tools/shard_job.py 20 8 $r 2>&1 | tail -1; done > $O/shard_jobs.txt 2>&1
timeout 60 tools/experiments/lds_occ.bin > $O/lds_granule.txt 2>&1
timeout 300 python tools/ws_mem.py > $O/ws_mem.txt 2>&1
# single-frame latency, ragged job
timeout 300 python tools/latency.py > $O/latency.txt 2>&1
timeout 600 python tools/ragged_bench.py > $O/ragged.jsonl 2>/dev/null
du -sh $O; head -14 $O/kernel_trace_one_lane_stats.txt | cut -c1-170; tail -1 $O/bench.json | cut -c1-600
