#!/bin/bash
# r06_t9: the global-pixel launch of a big pass on the lane's side stream only while no more than N lanes of the cascador are busy
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t9; mkdir -p $O; cd $R
run() { local label="$1"; shift; echo -n "$label: "; env "$@" PIPE_STEPS=120 python tools/pipe.py 2>>$O/log.txt | tail -1; }
for i in 1 2; do
for n in 8 1 2; do
run "side while <= $n busy, ahead2" PIPE_AHEAD=2 JDA_X_SIDE_BUSY_MAX=$n
run "side while <= $n busy, ahead1" PIPE_AHEAD=1 JDA_X_SIDE_BUSY_MAX=$n
done; done
for n in 8 1; do
echo "== bench, side while <= $n busy"
JDA_X_SIDE_BUSY_MAX=$n python bench.py --gpus 1 --steps 20 --warmup 5 2>>$O/log.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('step %.4f single %.4f onelane %.4f | cfg2 %.3f ms gpu %.3f sw %.3f | cpp %.2f | fddb %.3f cppjob %.2f pred8 %.2f | host %.3g pinned %.3g' % (d['ms_per_step'], c['single_caller_ms_per_step'], c['one_lane_ms_per_step'], c['config2_ms_per_call'], c['config2_gpu_ms_per_call'], c['config2_submit_wait_ms_per_call'], c['cpp_ms_per_step'], c['fddb_ms_per_job'], c['fddb_cpp_ms_per_job'], c['fddb_pred_speedup_8'], c['host_frames_windows_per_s'], c['host_frames_pinned_windows_per_s']))"
done
