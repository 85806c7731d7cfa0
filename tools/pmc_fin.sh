#!/bin/bash
# k_finish memory-path counters (one --pmc set per pass); output: gpurun_out/pmc_fin/*.txt
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc_fin; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export JDA_LANES=1 JDA_SIDE_STREAM=0 VAR_STEPS=5
rocprofv3 --list-avail 2>/dev/null | grep -E "^\s*(Name|name)\s*:|^gpu|Counter_Name|Name:" | sed 's/^\s*//' | sort -u | tr '\n' ' ' | head -c 20000 > $O/avail.txt
i=0
# (a fourth set -- TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum --
#  aborted rocprofv3 on this image and left it hanging: not collected)
for set in "TA_BUSY_avr TA_BUSY_max MemUnitBusy MemUnitStalled" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pf
  timeout 90 rocprofv3 --kernel-trace --pmc $set -d /tmp/pf -- python $R/tools/variants.py "" > /dev/null 2> $O/err_$i.txt
  db=$(find /tmp/pf -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db | grep -E "k_filter0|k_finish|k_scan<float, 4, false, 2" | cut -c1-150 > $O/set_$i.txt || echo "no db for: $set" > $O/set_$i.txt
done
cat $O/set_*.txt
