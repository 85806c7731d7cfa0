#!/usr/bin/env python3
"""Per-kernel sums of PMC counters from a rocprofv3 rocpd database:
   rocpd_pmc.py results.db [kernel-substring]  -> counter totals and per-dispatch averages"""
import sqlite3
import sys


def main(path, sub=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [d[1] for d in cur.execute('pragma table_info("counters_collection")')]
    rows = cur.execute("select * from counters_collection").fetchall()
    ki = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name")
    ci = cols.index("counter_name"); vi = cols.index("value")
    di = cols.index("dispatch_id")
    agg = {}
    for r in rows:
        if sub and sub not in r[ki]:
            continue
        key = (r[ki][:90], r[ci])
        a = agg.setdefault(key, [set(), 0.0])
        a[0].add(r[di]); a[1] += r[vi]
    for (k, c), (disp, tot) in sorted(agg.items()):
        print("%-14s dispatches %5d  total %16.1f  per-dispatch %14.2f  %s" % (c, len(disp), tot, tot / max(1, len(disp)), k))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
