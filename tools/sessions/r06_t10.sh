#!/bin/bash
# r06_t10: two tickets, no side stream next to another pass: slots of the persistent scan, bench --depth 2 / 3
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t10; mkdir -p $O; cd $R
run() { local label="$1"; shift; echo -n "$label: "; env "$@" PIPE_STEPS=120 python tools/pipe.py 2>>$O/log.txt | tail -1; }
for i in 1 2; do
run "ahead1 slots5" PIPE_AHEAD=1
run "ahead1 slots6" PIPE_AHEAD=1 JDA_SCAN_P_SLOTS=6
run "ahead1 slots4" PIPE_AHEAD=1 JDA_SCAN_P_SLOTS=4
run "ahead2 slots6" PIPE_AHEAD=2 JDA_SCAN_P_SLOTS=6
done
for d in 2 3 2 3; do
echo -n "bench --depth $d: "
python bench.py --gpus 1 --steps 20 --warmup 5 --depth $d 2>>$O/log.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('value %.4g step %.4f single %.4f | host %.3g pinned %.3g' % (d['value'], d['ms_per_step'], c['single_caller_ms_per_step'], c['host_frames_windows_per_s'], c['host_frames_pinned_windows_per_s']))"
done
