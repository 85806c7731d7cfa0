#!/bin/bash
# r06_t6: configs[2] with placed streams vs the runtime's deal; bench twice
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t6; mkdir -p $O; cd $R
for i in 1 2; do for p in 0 1; do echo -n "place$p: "; JDA_HWQ_PLACE=$p python tools/config2.py 2>>$O/log.txt | tail -2 | tr '\n' ' ' | cut -c1-260; echo; done; done
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 2>>$O/log.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('step %.4f | cfg2 %.3f ms gpu %.3f sw %.3f | cpp %.2f | fddb %.3f cppjob %.2f pred8 %.2f' % (d['ms_per_step'], c['config2_ms_per_call'], c['config2_gpu_ms_per_call'], c['config2_submit_wait_ms_per_call'], c['cpp_ms_per_step'], c['fddb_ms_per_job'], c['fddb_cpp_ms_per_job'], c['fddb_pred_speedup_8']))"; done
