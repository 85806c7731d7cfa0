mkdir -p gpurun_out/r06
for v in "A=0" "JDA_RAGGED_ONE_STREAM=-1" "JDA_RAGGED_ONE_STREAM=-1 JDA_RAGGED_CHUNK_WINDOWS_CPP=16000000" "JDA_RAGGED_CHUNK_WINDOWS=8000000" "A=1"; do
  echo "== $v"
  env $v python bench.py --no-cpu --no-x --no-config2 --no-allpass --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('value %.4g ms %.4f | fddb %.0f img/s %.3f ms pred8 %.2f | cpp %.4g w/s | fddb_cpp %.0f img/s %.2f ms' % (d['value'], d['ms_per_step'], c['fddb_images_per_s'], c['fddb_ms_per_job'], c['fddb_pred_speedup_8'], c['cpp_windows_per_s'], c['fddb_cpp_images_per_s'], c['fddb_cpp_ms_per_job']))"
done
