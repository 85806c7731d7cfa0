#!/bin/bash
# usage: tools/sweep.sh VAR v1 v2 ... ; prints scan/gpu/step ms for the cascade regime
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps 5 --warmup 1 --no-cpu --no-allpass 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); c=d['regimes']['cascade']
print('$VAR=$v', 'scan_ms %.3f gpu_ms %.3f call_ms %.3f step_ms %.3f host_ms %.3f win/s %.3e' % (c['scan_ms_per_step'], c['gpu_ms_per_step'], c.get('call_ms_per_step', 0), c['ms_per_step'], c['host_post_ms_per_step'], c['windows_per_s']))"
done
