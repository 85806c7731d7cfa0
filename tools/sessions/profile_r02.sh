#!/bin/bash
# Round-2 measurement set on the GPU box (outputs under gpurun_out/<tag>/, summaries copied to profiles/ by hand):
#   gpurun --timeout 1500 -- 'tools/sessions/profile_r02.sh r02_p'
TAG=${1:-r02_p}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --no-cpu --no-allpass > $O/bench_run2.json 2>> $O/bench.err
python bench.py --no-cpu --no-allpass > $O/bench_run3.json 2>> $O/bench.err
[ -x tools/lds_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/lds_bench tools/lds_bench.hip
tools/lds_bench > $O/lds_bench.txt 2>&1
cd /tmp; export TMPDIR=/tmp
V="python $R/tools/variants.py"
export VAR_STEPS=10
S="JDA_LANES=1 JDA_SIDE_STREAM=0"
# kernel traces: every launch of a step back to back on one stream / the default (two lanes)
env $S rocprofv3 --kernel-trace --stats -d $O/kt1 -- $V "" > $O/kt1_run.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt2 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --no-allpass > $O/kt2_bench.json 2>/dev/null
# HBM traffic: one counter per pass (13 passes of the batch each: 3 warm-up + 10)
env $S rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -- $V "" > /dev/null 2>&1
env $S rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -- $V "" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/cal_f -- python $R/tools/pmc_calib.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/cal_w -- python $R/tools/pmc_calib.py > /dev/null 2>&1
# SQ counters
env $S rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/sq1 -- $V "" > /dev/null 2>&1
env $S rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM -d $O/sq2 -- $V "" > /dev/null 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocpd_summary.py $(db kt1) k_ > $O/kernel_trace_one_lane_stats.txt
python tools/rocpd_summary.py $(db kt2) k_ > $O/kernel_trace_stats.txt
python tools/rocpd_pmc.py $(db pmc_f) > $O/pmc_hbm.txt; python tools/rocpd_pmc.py $(db pmc_w) >> $O/pmc_hbm.txt
python tools/rocpd_pmc.py $(db sq1) > $O/pmc_sq.txt; python tools/rocpd_pmc.py $(db sq2) >> $O/pmc_sq.txt
python tools/pmc_traffic.py $(db cal_f) $(db cal_w) $(db pmc_f) $(db pmc_w) 13 "$TAG: JDA_LANES=1 JDA_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python tools/variants.py ''" > $O/hbm_traffic.json
# the other configs, latency, host frames
python tests/gpu_configs.py both > $O/other_configs.txt 2>&1
python tools/latency.py >> $O/other_configs.txt 2>&1
python tools/host_batch.py 256 >> $O/other_configs.txt 2>&1
python tools/fddb_bench.py --threads 8 --dialect c > $O/fddb_shaped.jsonl 2>/dev/null
find $O -name "*.db" -delete; rm -rf $O/kt1 $O/kt2 $O/pmc_f $O/pmc_w $O/cal_f $O/cal_w $O/sq1 $O/sq2
du -sh $O; head -12 $O/kernel_trace_one_lane_stats.txt | cut -c1-170; cat $O/hbm_traffic.json | head -40
