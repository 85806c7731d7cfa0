#!/bin/bash
cd "$(dirname "$0")/.."
for v in "$@"; do echo "== $v"; env $v python tools/ragged_bench.py --variants device --reps 7 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-8s %.0f images/s %.2f ms (best %.2f)  %.3e windows/s' % (d['variant'], d['value'], d['ms_per_job'], d['best_ms'], d['windows_per_s']))"; done
