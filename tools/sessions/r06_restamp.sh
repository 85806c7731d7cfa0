#!/bin/bash
# the counter passes behind profiles/hbm_traffic.json and the two x_allpass JSONs again, on the final device code (k_finish.hip changed in its
# similarity-transform instantiations: the hash the JSONs carry has to be this code's), then the driver's bench line
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
bash tools/traffic_refresh.sh r06_restamp > /dev/null 2>&1
bash tools/sessions/r06_x.sh > /dev/null 2>&1
cp gpurun_out/r06_x/x_allpass_traffic.json gpurun_out/r06_x/x_allpass_T14_traffic.json profiles/
O=$R/gpurun_out/r06_restamp
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err
python bench.py > $O/bench_default.json 2>> $O/bench.err
python - <<'P'
import json
for f in ("bench_driver_args.json", "bench_default.json"):
    d = json.loads(open("gpurun_out/r06_restamp/" + f).read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
    print(f, "steps", d["steps"], "value %.4g ms %.4f | traffic ok %s %s %s | parity %s | frac %.4f hbm %.3f/%.3f T14 %.3f/%.3f" % (d["value"], d["ms_per_step"], r.get("traffic_from_this_device_code"), r.get("hbm_regime_traffic_from_this_device_code"), d.get("roofline_hbm_regime_beyond_mall", {}).get("traffic_from_this_device_code"), c.get("parity_checked"), r["frac"], r["hbm_regime_frac"], r["hbm_regime_frac_weight_rows"], r["hbm_regime_beyond_mall_frac"], r["hbm_regime_beyond_mall_frac_weight_rows"]))
P
