#!/bin/bash
# r06_t5: GPU suite + bench on the library with placed streams
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t5; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/suite.txt 2>&1; grep -E "passed|failed|error" $O/suite.txt | tail -3
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'P'
import json
d = json.loads(open("gpurun_out/r06_t5/bench.json").read().strip().splitlines()[-1]); c = d["config"]
print("value %.4g ms %.4f | single %.4f | fddb %.0f (%.3f ms) pred8 %.2f | cpp %.3g fddb_cpp %.0f (%.2f ms) | host %.3g pinned %.3g | config2 %.3g | frac %.4f kernel_ms %.4f | parity %s" % (
  d["value"], d["ms_per_step"], c["single_caller_ms_per_step"], c["fddb_images_per_s"], c["fddb_ms_per_job"], c["fddb_pred_speedup_8"], c["cpp_windows_per_s"], c["fddb_cpp_images_per_s"], c["fddb_cpp_ms_per_job"],
  c["host_frames_windows_per_s"], c["host_frames_pinned_windows_per_s"], c["config2_windows_per_s"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], c["parity_checked"]))
P
