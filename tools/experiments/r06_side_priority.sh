# r06: the lane's side stream with a priority of its own (a hardware queue of its own) -- bench context, where the process has many streams
for v in "A=0" "JDA_SIDE_PRIORITY=1" "JDA_SIDE_PRIORITY=-1" "JDA_RAGGED_SINGLE_WINDOWS=0" "A=1" "JDA_SIDE_PRIORITY=1"; do
  echo "=== $v"
  env $v python bench.py --no-cpu --no-x --no-config2 --no-allpass --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; p=d['fddb']['predicted_strong_scaling']['8']
print('value %.4g ms %.4f single %.4f | fddb %.0f img/s %.3f ms pred8 %.2f shard min %.3f max %.3f | fddb_cpp %.0f img/s' % (d['value'], d['ms_per_step'], c['single_caller_ms_per_step'], c['fddb_images_per_s'], c['fddb_ms_per_job'], c['fddb_pred_speedup_8'], p['min_shard_ms'], p['max_shard_ms'], c['fddb_cpp_images_per_s']))"
done
