#!/bin/bash
# headline leg (submit/wait, two batches in flight, 4 rotating batches) under several settings of the experiment knobs
#   tools/bench_variants.sh "NAME=V NAME=V" "NAME=V" ...
for spec in "$@"; do
  env $spec python bench.py --steps 60 --warmup 6 --no-cpu --no-allpass 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); c=d['regimes']['cascade']; s=d['regimes']['cascade_single_caller']
print('%-60s step %.3f ms  %.3e win/s   single caller %.3f ms' % ('''$spec''' or '(defaults)', c['ms_per_step'], c['windows_per_s'], s['ms_per_step']))"
done
