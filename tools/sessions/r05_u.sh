#!/bin/bash
# r05_u: final check of the tree: GPU suite, smoke, the driver's bench invocation
mkdir -p gpurun_out/r05_u
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05_u/suite.txt
cat gpurun_out/r05_u/suite.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r05_u/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_u/bench.json 2> gpurun_out/r05_u/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05_u/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('fddb_images_per_s'), d['config'].get('config2_windows_per_s'), d['roofline'].get('traffic_from_this_device_code'))
P
