#!/bin/bash
# Round 5, first contact: the GPU suite, then the default bench line.   gpurun --timeout 1500 -- 'tools/sessions/r05_a.sh r05_a'
TAG=${1:-r05_a}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
tail -3 $O/bench.time; tail -c 1500 $O/bench.err
python - <<EOF
import json
try:
    l = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("value %.4g ms %.4f" % (l["value"], l["ms_per_step"]))
    print({k: v for k, v in l["config"].items() if k.startswith(("fddb", "config2", "config4", "allpass", "single", "dist"))})
    print({k: v for k, v in l["roofline"].items() if not isinstance(v, str)})
except Exception as e:
    print("bench line unreadable", e)
EOF
