#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into text:
per-kernel calls/total/avg/min/max, and (optionally) per-dispatch rows of one kernel."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name if len(name) < 150 else name[:147] + "..."


def main(path, detail=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [d[1] for d in cur.execute('pragma table_info("kernels")')]
    rows = cur.execute("select * from kernels").fetchall()
    ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
    gi = [cols.index(c) for c in ("grid_x", "grid_y", "grid_z")] if "grid_x" in cols else None
    wi = cols.index("workgroup_x") if "workgroup_x" in cols else None
    li = cols.index("lds_size") if "lds_size" in cols else None
    agg = {}
    for r in rows:
        a = agg.setdefault(r[ni], [0, 0, 1 << 62, 0])
        d = r[ei] - r[si]
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    print("%-8s %12s %10s %10s %10s %6s  %s" % ("calls", "total_us", "avg_us", "min_us", "max_us", "pct", "kernel"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-8d %12.1f %10.2f %10.2f %10.2f %6.2f  %s" % (a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3,
                                                             100.0 * a[1] / tot, short(name)))
    if detail:
        print("\nper-dispatch rows matching %r (columns: %s)" % (detail, cols))
        seen = 0
        for r in rows:
            if detail in r[ni]:
                extra = ""
                if gi:
                    extra = " grid=%s wg=%s lds=%s" % ([r[i] for i in gi], r[wi] if wi is not None else "?", r[li] if li is not None else "?")
                print("%10.2f us%s  %s" % ((r[ei] - r[si]) / 1e3, extra, short(r[ni])[:90]))
                seen += 1
                if seen >= 60:
                    break


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
