# r06: k_finish's regression, 64 weight rows in flight per batch (fp32 default now) against 32 (libjda_rb32.so = -DFIN_ROW_BATCH=32)
for lib in "" "$PWD/jda_amd/libjda_rb32.so"; do
  echo "=== ${lib:-product (fp32: 64 rows per batch)}"
  export JDA_LIB_PATH=$lib; [ -z "$lib" ] && unset JDA_LIB_PATH
  python tools/shard_job.py 20 | tail -1
  python tools/x_allpass.py --steps 1 2>&1 | tail -2 | head -1
  python bench.py --no-cpu --no-x --no-config2 --no-allpass --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('value %.4g ms %.4f single %.4f | fddb %.0f img/s %.3f ms pred8 %.2f | fddb_cpp %.0f img/s' % (d['value'], d['ms_per_step'], c['single_caller_ms_per_step'], c['fddb_images_per_s'], c['fddb_ms_per_job'], c['fddb_pred_speedup_8'], c['fddb_cpp_images_per_s']))"
done
