# r06: dialect-CPP legs in bench.py's process under a few settings
for v in "A=0" "JDA_RAGGED_LANES=2" "JDA_RAGGED_LANES=4" "JDA_FIRST_PHASE=32" "JDA_FILTER0=0" "JDA_RAGGED_CHUNK_WINDOWS_CPP=12000000" "JDA_RAGGED_CHUNK_WINDOWS_CPP=6000000" "A=1"; do
  echo "=== $v"
  env $v python bench.py --no-cpu --no-x --no-config2 --no-allpass --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('cpp %.4g w/s %.2f ms | fddb_cpp %.0f img/s %.2f ms | value %.4g' % (c['cpp_windows_per_s'], c['cpp_ms_per_step'], c['fddb_cpp_images_per_s'], c['fddb_cpp_ms_per_job'], d['value']))"
done
