#!/usr/bin/env python3
"""profiles/r03_x_allpass.txt: the HBM-regime line of BASELINE.json configs[4] made checkable.
   x_traffic.py <gather_calib.db> <x_allpass.db> <x_allpass_stdout.txt> [<gather_calib_rdreq.db> <x_allpass_rdreq.db>]
FETCH_SIZE (KiB) of the gather calibration (tools/pmc_calib.py gather: a known number of 544-byte rows) gives the
counter's bytes per USEFUL byte for this access pattern on the cache-resident table and on a 1 GiB table; applied to the
dominant k_finish dispatch of tools/x_allpass.py it gives that launch's fabric-side traffic in the same units."""
import re, sqlite3, sys


def per_dispatch(path, counter, sub):
    cur = sqlite3.connect(path).cursor()
    cols = [d[1] for d in cur.execute('pragma table_info("counters_collection")')]
    ki = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name")
    ci, vi, di = cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
    out = {}
    for r in cur.execute("select * from counters_collection"):
        if r[ci] == counter and sub in r[ki]:
            out[r[di]] = out.get(r[di], 0.0) + r[vi]
    return [out[k] for k in sorted(out)]


def durations(path, sub):
    cur = sqlite3.connect(path).cursor()
    cols = [d[1] for d in cur.execute('pragma table_info("kernels")')]
    ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
    return [(r[ei] - r[si]) * 1e-9 for r in cur.execute("select * from kernels") if sub in r[ni]]


cal, run, txt = sys.argv[1:4]
useful = 30000 * 2000 * 136 * 4
g = per_dispatch(cal, "FETCH_SIZE", "calib_gather_rows")          # 2 dispatches per table: [cache-resident x2, 1 GiB x2]
gd = durations(cal, "calib_gather_rows")
print("# gather calibration (tools/pmc_calib.hip calib_gather_rows): %d waves x 2000 rows of 544 B = %.2f GB useful per dispatch" % (30000, useful / 1e9))
f = []
for name, i in (("W-sized table, 243.7 MB (inside the 256 MB Infinity Cache)", 1), ("1 GiB table (4x the Infinity Cache)", 3)):
    fac = g[i] * 1024.0 / useful
    f.append(fac)
    print("  %-62s FETCH_SIZE %.4e KiB  = %.3f counted bytes per useful byte; %.1f ms -> %.2f TB/s useful" % (name, g[i], fac, gd[i] * 1e3, useful / gd[i] / 1e12))
xs = per_dispatch(run, "FETCH_SIZE", "k_finish")
xd = durations(run, "k_finish")
big = max(range(len(xs)), key=lambda i: xs[i])
m = re.search(r"(\d+) windows", open(txt).read())
win = int(m.group(1)) if m else 303222
mt = re.search(r"all-pass T=(\d+)", open(txt).read())
T, K, L, D = (int(mt.group(1)) if mt else 7), 2000, 68, 6
rows = win * T * K * 2 * L * 4
alg = win * T * K * ((D - 1) * 34 + 16) + rows + win * 2 * L * 4
print("# tools/x_allpass.py, the k_finish dispatch that walks the stages (%d windows x %d carts):" % (win, T * K))
print("  FETCH_SIZE %.4e KiB = %.3f TB as counted; duration %.1f ms" % (xs[big], xs[big] * 1024 / 1e12, xd[big] * 1e3))
print("  algorithmic bytes (SURVEY 8d): %.3f TB (weight rows %.3f TB) -> %.2f TB/s = %.1f %% of the 8 TB/s HBM peak" % (alg / 1e12, rows / 1e12, alg / xd[big] / 1e12, 100 * alg / xd[big] / 8e12))
for name, fac in (("cache-resident gather factor", f[0]), ("1 GiB gather factor", f[1])):
    tr = xs[big] * 1024 / fac
    print("  traffic calibrated with the %s (%.3f): %.3f TB useful-equivalent = %.2f x the weight rows, %.2f x the algorithmic bytes; %.2f TB/s"
          % (name, fac, tr / 1e12, tr / rows, tr / alg, tr / xd[big] / 1e12))
if len(sys.argv) >= 6:
    for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_DRAM_sum"):
        a = per_dispatch(sys.argv[4], c, "calib_gather_rows"); b = per_dispatch(sys.argv[5], c, "k_finish")
        if a and b:
            print("  %-24s gather calib (cache-resident / 1 GiB) %.4e / %.4e   k_finish %.4e" % (c, a[1], a[3] if len(a) > 3 else float('nan'), max(b)))
import json, os
rd = None
if len(sys.argv) >= 6:
    b = per_dispatch(sys.argv[5], "TCC_EA0_RDREQ_sum", "k_finish")
    rd = max(b) if b else None
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_traffic
rec = {"source": os.environ.get("JDA_X_SOURCE", "tools/sessions/ARCHIVE_r02_r04.txt (r03_x.sh)") + ": rocprofv3 --kernel-trace --pmc FETCH_SIZE (and TCC_EA0_RDREQ_sum) -- python tools/x_allpass.py; calibration tools/pmc_calib.py gather",
       "kernel_sources_sha256": pmc_traffic.kernel_sources_sha256(), "stages": T, "w_bytes": T * K * 32 * 2 * L * 4,
       "windows": win, "duration_s": xd[big], "algorithmic_bytes": alg, "weight_row_bytes": rows,
       "fetch_size_bytes_as_counted": xs[big] * 1024, "gather_calibration_counted_per_useful_byte": {"w_sized_table": f[0], "1GiB_table": f[1]},
       "traffic_useful_equivalent_bytes": xs[big] * 1024 / f[0],
       "fabric_read_requests": rd, "traffic_line_bytes": (rd * 128 if rd else xs[big] * 1024 * 2),
       "note": "FETCH_SIZE tallies a 128-byte fabric request as 64 bytes on gfx950 (request count x 64 B = FETCH_SIZE, checked here); "
               "traffic_line_bytes = requests x 128 B; Infinity-Cache hits are counted, no counter separates them from HBM reads"}
json.dump(rec, open(os.path.join(os.path.dirname(os.path.abspath(txt)), os.environ.get("JDA_X_JSON", "x_allpass_traffic.json")), "w"), indent=1)
print("# no rocprofv3 counter on this part separates Infinity-Cache hits from HBM reads: TCC_EA0_RDREQ_DRAM counts L2 requests routed to")
print("# local memory, Infinity-Cache hits included (MI355X_MICROARCH.md, HBM); the 1 GiB row shows what the same gather costs when it cannot hit.")
