#!/bin/bash
# fddb.predicted_strong_scaling of the bench under the given environments
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; TAG=${TAG:-fddb_scaling}
for v in "$@"; do
  echo "=== $v"
  env $v timeout 300 python bench.py --no-cpu --no-allpass --no-x --steps 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); f = d['fddb']
        print('job %.3f ms  host job %.0f img/s ' % (f['ms_per_job'], f.get('host_images_per_s', 0)), {k: (round(v['max_shard_ms'], 3), round(v['speedup'], 2)) for k, v in f['predicted_strong_scaling'].items()})
"
done > gpurun_out/$TAG.log 2>&1
cat gpurun_out/$TAG.log
