#!/bin/bash
# final bench lines with two batches in flight (bench.py's default since session r06_t15)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t16; mkdir -p $O; cd $R
python bench.py > $O/bench.json 2> $O/bench.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2>> $O/bench.err ) 2>&1 | grep real
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-allpass --no-x --no-config2 > $O/bench_driver_args_2.json 2>> $O/bench.err
python - <<'P'
import json
for f in ("bench.json", "bench_driver_args.json", "bench_driver_args_2.json"):
    d = json.loads(open("gpurun_out/r06_t16/" + f).read().strip().splitlines()[-1]); c = d["config"]
    print(f, "steps", d["steps"], "value %.4g ms %.4f in flight %s | fddb %.0f pred8 %s | cppjob %s | host %.3g | traffic ok %s parity %s" % (d["value"], d["ms_per_step"], c["batches_in_flight_per_gpu"], c["fddb_images_per_s"], c.get("fddb_pred_speedup_8"), c.get("fddb_cpp_images_per_s"), c["host_frames_windows_per_s"], d["roofline"].get("traffic_from_this_device_code"), c.get("parity_checked")))
P
