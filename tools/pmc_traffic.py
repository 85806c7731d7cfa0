#!/usr/bin/env python3
"""profiles/hbm_traffic.json from rocprofv3 PMC passes (one counter per pass, as /opt/skills/guides/MI355X_MICROARCH.md
prescribes):   tools/pmc_traffic.py calib_f.db calib_w.db run_f.db run_w.db passes_of_the_batch tag

FETCH_SIZE / WRITE_SIZE are in KiB.  The gfx950 correction (FETCH_SIZE tallies 128-B requests as 64 B) is not
assumed: the factors are measured by tools/pmc_calib.hip on a 1 GiB buffer in the kernels' own access widths.
Kernels are grouped by pixel mode: k_scan<..., 1|3, ...> = LDS-tiled launches, k_scan<..., 2, ...> = global-pixel."""
import json
import re
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


def group(name):
    m = re.search(r"k_scan<\w+, \d+, \w+, (\d), \d+(?:, \w+)*>", name)
    if m:
        return "k_scan_global" if m.group(1) == "2" else "k_scan_lds"
    if "k_scan_p<" in name:          # the persistent form of the LDS-tiled scan
        return "k_scan_lds"
    for k in ("k_filter0", "k_finish_wide", "k_finish", "k_stage", "k_prep_stage0", "k_enqueue", "k_resize"):
        if k in name:
            return k
    return None


DEVICE_HEADERS = ("kernels.h", "kernels_common.h", "finish_common.h", "scan_walk.h", "k_scan_impl.h")


def kernel_sources_sha256():
    """Hash of the device code the counters were taken on (csrc/*.hip and the headers they include; the host-side
    headers of the same directory are not device code): bench.py compares it with the sources it runs on, so a traffic
    figure echoed from a stale file is flagged in the line."""
    import glob, hashlib, os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jda_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + [os.path.join(root, x) for x in DEVICE_HEADERS]):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def main(cf, cw, bf, bw, passes, tag, calib_bytes=1 << 30):
    passes = int(passes)
    f, wv = sums(cf, "FETCH_SIZE"), sums(cw, "WRITE_SIZE")
    pick = lambda d, sub: [sum(x) for x in zip(*[(c, v) for k, (c, v) in d.items() if sub in k])] or [0, 0.0]
    n4, r4 = pick(f, "calib_read4"); n16, r16 = pick(f, "calib_read16"); nw, w4 = pick(wv, "calib_write4")
    f4 = calib_bytes / (r4 / n4 * 1024.0); f16 = calib_bytes / (r16 / n16 * 1024.0); wf = calib_bytes / (w4 / nw * 1024.0)
    out = {"source": tag, "kernel_sources_sha256": kernel_sources_sha256(), "unit": "bytes per step (one pass over the 256-frame batch)", "passes_profiled": passes,
           "calibration": {"buffer_bytes": calib_bytes, "fetch_factor_4B_per_lane": f4, "fetch_factor_16B_per_lane": f16,
                           "write_factor": wf}, "kernels": {}}
    bfetch, bwrite = sums(bf, "FETCH_SIZE"), sums(bw, "WRITE_SIZE")
    for src, key, fac in ((bfetch, "hbm_read_bytes", f16), (bwrite, "hbm_write_bytes", wf)):
        for name, (cnt, v) in src.items():
            g = group(name)
            if g is None:
                continue
            e = out["kernels"].setdefault(g, {"dispatches_per_step": 0.0, "hbm_read_bytes": 0.0, "hbm_write_bytes": 0.0})
            e[key] += v * 1024.0 * fac / passes
            if key == "hbm_read_bytes":
                e["dispatches_per_step"] += cnt / passes
    k = out["kernels"]
    out["k_scan_lds_bytes_per_step"] = k.get("k_scan_lds", {}).get("hbm_read_bytes", 0) + k.get("k_scan_lds", {}).get("hbm_write_bytes", 0)
    out["k_scan_bytes_per_step"] = out["k_scan_lds_bytes_per_step"] + k.get("k_scan_global", {}).get("hbm_read_bytes", 0) + \
        k.get("k_scan_global", {}).get("hbm_write_bytes", 0)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
