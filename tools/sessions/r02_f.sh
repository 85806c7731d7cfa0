#!/bin/bash
O=gpurun_out/r02_f; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/tests.txt
python bench.py --no-cpu > $O/bench_nocpu.json 2> $O/bench.err
S="JDA_LANES=1 JDA_SIDE_STREAM=0"
python tools/variants.py "" "$S" > $O/variants.txt 2>&1
python tools/latency.py > $O/latency.txt 2>&1
cat $O/tests.txt $O/variants.txt $O/latency.txt; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r02_f/bench_nocpu.json"))
print({k:d[k] for k in ("value","ms_per_step")}); print(d["config"]); print({k:v for k,v in d["roofline"].items() if k not in ("what","measured_with","hbm_side","traffic_source")})
for k,v in d["regimes"].items():
    if v and "ms_per_step" in v: print(k, {x:v[x] for x in ("ms_per_step","gpu_ms_per_step","scan_ms_per_step","scan_lds_ms_per_step","windows_per_s")})
print(d["regimes"]["host_frames"])
PY
