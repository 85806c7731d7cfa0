#!/usr/bin/env python3
"""Steady-state window of a kernel trace of tools/pipe.py (the bench's headline leg): (start, end, duration, queue, kernel) for ~4 steps in the middle of
the timed region, and per queue the busy time.   python tools/bench_timeline.py DB"""
import re, sqlite3, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"jda::", "", n)
    m = re.match(r"void (\w+)<(.*)>\(", n)
    if not m:
        return re.sub(r"\(.*", "", n)[:40]
    args = [a.strip() for a in m.group(2).split(",")]
    if m.group(1) == "k_scan":      # Real, DEPTH, TRACE, MODE, BLOCK, ...
        return "k_scan mode%s b%s" % (args[3], args[4])
    return m.group(1)


def main(path):
    db = sqlite3.connect(path); cur = db.cursor()
    kc = [d[1] for d in cur.execute('pragma table_info("kernels")')]
    qcol = "queue_id" if "queue_id" in kc else ("stream_id" if "stream_id" in kc else None)
    rows = list(cur.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")))
    # (tools/pipe.py runs the leg twice: the window sits in the middle of the second run)
    fin = [r for r in rows if "k_finish<" in r[0]]
    mid = fin[(3 * len(fin)) // 4][1]
    t0, t1 = mid - 3.2e6, mid + 3.2e6
    busy = {}
    print("%10s %10s %8s  %-5s %s" % ("start_ms", "end_ms", "ms", "queue", "kernel"))
    for r in rows:
        if r[2] < t0 or r[1] > t1:
            continue
        q = r[3] if qcol else 0
        busy[q] = busy.get(q, 0) + (r[2] - r[1])
        print("%10.3f %10.3f %8.3f  q%-4s %s" % ((r[1] - t0) / 1e6, (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e6, q, short(r[0])))
    print("kernel time per queue in the window (ms):", {k: round(v / 1e6, 3) for k, v in busy.items()}, "window 6.4 ms")


if __name__ == "__main__":
    main(sys.argv[1])
