#!/usr/bin/env python3
"""Timeline of a host-frame run from a rocprofv3 rocpd database taken with --kernel-trace --memory-copy-trace:
every copy >= 1 MB and every kernel, as (start, end, queue/stream, what), for a window of the run -- to see where a
batch waits.   python tools/host_timeline.py DB [from_ms to_ms]"""
import re, sqlite3, sys


def main(path, t_from=None, t_to=None):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    ev = []
    kc = [d[1] for d in cur.execute('pragma table_info("kernels")')]
    qcol = "queue_id" if "queue_id" in kc else ("stream_id" if "stream_id" in kc else None)
    for r in cur.execute("select name, start, end%s from kernels" % ((", " + qcol) if qcol else "")):
        nm = re.sub(r"\(anonymous namespace\)::", "", r[0]); nm = re.sub(r"<.*", "", nm)[:28]
        ev.append((r[1], r[2], "q%s" % (r[3] if qcol else "?"), nm))
    mt = [t for t in tabs if "memory_cop" in t.lower()]
    for t in mt[:1]:
        cols = [d[1] for d in cur.execute('pragma table_info("%s")' % t)]
        print("copy table %s: %s" % (t, cols), file=sys.stderr)
        si, ei = cols.index("start"), cols.index("end")
        zi = cols.index("size") if "size" in cols else None
        ni = cols.index("name") if "name" in cols else None
        for r in cur.execute('select * from "%s"' % t):
            sz = r[zi] if zi is not None else 0
            ev.append((r[si], r[ei], "copy", "%s %.2f MB" % (r[ni] if ni is not None else "", (sz or 0) / 1e6)))
    ev.sort()
    t0 = ev[0][0]
    for s, e, q, nm in ev:
        a, b = (s - t0) / 1e6, (e - t0) / 1e6
        if t_from is not None and (b < t_from or a > t_to):
            continue
        if q == "copy" and "0.00 MB" in nm or "0.01 MB" in nm:
            tag = "   ."
        else:
            tag = ""
        print("%10.3f %10.3f %8.3f  %-6s %s%s" % (a, b, b - a, q, nm, tag))


if __name__ == "__main__":
    main(sys.argv[1], *(float(x) for x in sys.argv[2:4]))
