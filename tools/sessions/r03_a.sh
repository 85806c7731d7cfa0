#!/bin/bash
# round 3, session a: GPU suite, bench line, single-frame latency with and without the predicted finishing launches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03a_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r03a_pytest.txt
tail -5 gpurun_out/r03a_pytest.txt
timeout 600 python bench.py --steps 100 --warmup 8 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03a_bench.json').read().strip().splitlines()[-1])
print("value %.3e ms/step %.3f single %.3f ms host pinned %.3e pageable %.3e roofline frac %.3f lds ms %.3f" % (d["value"], d["ms_per_step"], d["config"]["single_caller_ms_per_step"], d["config"]["host_frames_pinned_windows_per_s"], d["config"]["host_frames_windows_per_s"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"]))
PY
timeout 300 python tools/latency.py > gpurun_out/r03a_latency.txt 2>&1
JDA_PREDICT=0 timeout 300 python tools/latency.py >> gpurun_out/r03a_latency.txt 2>&1
cat gpurun_out/r03a_latency.txt
