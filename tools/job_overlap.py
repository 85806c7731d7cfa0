"""How the kernels of the LAST job in a rocprofv3 kernel-trace database overlap: union of the kernel intervals (time the GPU
ran anything), sum of the durations, the longest gaps, and the same per queue.   python tools/job_overlap.py <db> [first-kernel pattern]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "k_repack"
cols = [d[1] for d in cur.execute('pragma table_info("kernels")')]
rows = cur.execute("select * from kernels").fetchall()
ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
qi = cols.index("queue_id") if "queue_id" in cols else None
rows.sort(key=lambda r: r[si])
starts = [i for i, r in enumerate(rows) if pat in r[ni]]
# jobs are separated by an idle gap of more than 1 ms in front of a `pat` kernel
first = starts[-1]
for i in reversed(starts):
    prev_end = max(r[ei] for r in rows[:i]) if i else 0
    first = i
    if i == 0 or rows[i][si] - prev_end > 1_000_000: break
job = rows[first:]
t0, t1 = job[0][si], max(r[ei] for r in job)
iv = sorted((r[si], r[ei]) for r in job)
union, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e in iv[1:]:
    if s > cur_e: union += cur_e - cur_s; gaps.append((s - cur_e, cur_e - t0)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
union += cur_e - cur_s
tot = sum(r[ei] - r[si] for r in job)
print("last job: %d dispatches, span %.3f ms, GPU busy (union) %.3f ms, sum of kernel durations %.3f ms (overlap factor %.2f)" % (len(job), (t1 - t0) / 1e6, union / 1e6, tot / 1e6, tot / union))
print("longest gaps (ms, at ms):", [(round(g / 1e6, 3), round(a / 1e6, 3)) for g, a in sorted(gaps, reverse=True)[:6]])
by = {}
for r in job:
    nm = re.sub(r"^void jda::|^jda::", "", r[ni]).split("(")[0][:40]
    by.setdefault(nm, [0, 0]); by[nm][0] += 1; by[nm][1] += r[ei] - r[si]
for nm, (n, d) in sorted(by.items(), key=lambda kv: -kv[1][1]): print("  %-42s %4d x  %.3f ms" % (nm, n, d / 1e6))
if qi is not None:
    for q in sorted(set(r[qi] for r in job)):
        rq = [r for r in job if r[qi] == q]
        print("  queue %s: %d dispatches, busy %.3f ms, first start %.3f, last end %.3f" % (q, len(rq), sum(r[ei] - r[si] for r in rq) / 1e6, (rq[0][si] - t0) / 1e6, (max(r[ei] for r in rq) - t0) / 1e6))
