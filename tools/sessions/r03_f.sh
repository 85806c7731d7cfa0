#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03f_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r03f_pytest.txt
tail -4 gpurun_out/r03f_pytest.txt
timeout 300 python tools/latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03f_latency.txt
JDA_LIB_PATH=$PWD/jda_amd/libjda_timing.so timeout 300 python tools/wide_timing.py 2>&1 | grep -v amdgpu.ids | tail -12
timeout 600 python bench.py --steps 100 --warmup 8 > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03f_bench.json').read().strip().splitlines()[-1])
print("value %.3e ms/step %.3f single %.3f ms host pinned %.3e pageable %.3e roofline frac %.3f lds ms %.3f fddb %.0f img/s (host %.0f)" % (d["value"], d["ms_per_step"], d["config"]["single_caller_ms_per_step"], d["config"]["host_frames_pinned_windows_per_s"], d["config"]["host_frames_windows_per_s"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["fddb_images_per_s"], d["fddb"].get("host_images_per_s", 0)))
print("gpu_ms", d["regimes"]["cascade_one_lane"]["gpu_ms_per_step"], "scan", d["regimes"]["cascade_one_lane"]["scan_ms_per_step"])
PY
