#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the rocprofv3 PMC passes (separate passes for
FETCH_SIZE and WRITE_SIZE, as guides/MI355X_MICROARCH.md prescribes):

  tools/make_traffic_json.py calib_f.db calib_w.db bench_f.db bench_w.db steps_in_profiled_run

FETCH_SIZE / WRITE_SIZE are in KiB.  The gfx950 correction (FETCH_SIZE counts 128-B
requests as 64 B) is not assumed: the factor is measured by the calibration kernels
of tools/pmc_calib.hip on a 1 GiB buffer in k_scan's own 4 B/lane access pattern."""
import json
import sqlite3
import sys


def sums(path, counter):
    cur = sqlite3.connect(path).cursor()
    cols = [d[1] for d in cur.execute('pragma table_info("counters_collection")')]
    ki = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name")
    ci, vi, di = cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
    out = {}
    for r in cur.execute("select * from counters_collection"):
        if r[ci] != counter:
            continue
        a = out.setdefault(r[ki], [set(), 0.0])
        a[0].add(r[di]); a[1] += r[vi]
    return {k: (len(v[0]), v[1]) for k, v in out.items()}


def pick(d, sub):
    n = tot = 0
    for k, (cnt, v) in d.items():
        if sub in k:
            n += cnt; tot += v
    return n, tot


def main(cf, cw, bf, bw, steps, calib_bytes=1 << 30):
    steps = int(steps)
    f, wv = sums(cf, "FETCH_SIZE"), sums(cw, "WRITE_SIZE")
    n4, r4 = pick(f, "calib_read4"); n16, r16 = pick(f, "calib_read16"); nw, w4 = pick(wv, "calib_write4")
    fetch_factor4 = calib_bytes / (r4 / n4 * 1024.0)
    fetch_factor16 = calib_bytes / (r16 / n16 * 1024.0)
    write_factor = calib_bytes / (w4 / nw * 1024.0)
    bfetch, bwrite = sums(bf, "FETCH_SIZE"), sums(bw, "WRITE_SIZE")
    out = {"unit": "bytes per step (one pass over the 256-frame batch)", "steps_profiled": steps,
           "calibration": {"buffer_bytes": calib_bytes, "fetch_factor_4B_per_lane": fetch_factor4,
                           "fetch_factor_16B_per_lane": fetch_factor16, "write_factor": write_factor},
           "kernels": {}}
    for name in ("k_scan", "k_finish"):
        nf, tf = pick(bfetch, name); nw_, tw = pick(bwrite, name)
        # the profiled run has steps + warmup passes; normalise by the dispatch count instead
        per_step_disp = nf / max(1, (steps + 1))
        fb = tf * 1024.0 * fetch_factor4 / (steps + 1)
        wb = tw * 1024.0 * write_factor / (steps + 1)
        out["kernels"][name] = {"dispatches_per_step": per_step_disp, "hbm_read_bytes": fb, "hbm_write_bytes": wb}
    out["k_scan_bytes_per_step"] = out["kernels"]["k_scan"]["hbm_read_bytes"] + out["kernels"]["k_scan"]["hbm_write_bytes"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
