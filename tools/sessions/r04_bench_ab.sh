#!/bin/bash
# headline leg of bench.py (submit/wait, two batches in flight) under the given environments, alternating twice
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r04_bench_ab}
cd $R
for rep in 1 2; do
for v in "$@"; do
  echo "=== $v"
  env $v timeout 300 python bench.py --no-cpu --no-allpass --no-x --steps 60 $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('value %.4g  ms_per_step %.4f  fddb %s' % (d['value'], d['ms_per_step'], d.get('config', {}).get('fddb_images_per_s')))
"
done
done > $R/gpurun_out/$TAG.log 2>&1
cat $R/gpurun_out/$TAG.log
