#!/bin/bash
# dialect-CPP FDDB-shaped job inside bench.py's process (its stream / queue context), chunk-size sweep
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
run() { env "$@" python bench.py --no-cpu --no-allpass --no-x --no-config2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('%-50s cpp job %.2f ms %.0f img/s | C job %.3f ms pred8 %.2f max shard %.3f | step %.4f' % ('$*', c['fddb_cpp_ms_per_job'], c['fddb_cpp_images_per_s'], c['fddb_ms_per_job'], c['fddb_pred_speedup_8'], d['fddb']['predicted_strong_scaling']['8']['max_shard_ms'], d['ms_per_step']))"; }
for i in 1 2; do
run A=0
run JDA_RAGGED_CHUNK_WINDOWS_CPP=6000000
run JDA_RAGGED_CHUNK_WINDOWS_CPP=10000000
run JDA_RAGGED_CHUNK_WINDOWS_CPP=12000000
run JDA_RAGGED_CHUNK_WINDOWS_CPP=16000000
done
