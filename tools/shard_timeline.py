"""Timeline of the LAST job in a rocprofv3 kernel-trace database of tools/shard_job.py: start offset, duration, kernel, grid."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [d[1] for d in cur.execute('pragma table_info("kernels")')]
rows = cur.execute("select * from kernels").fetchall()
ni, si, ei, gx, wx = cols.index("name"), cols.index("start"), cols.index("end"), cols.index("grid_x"), cols.index("workgroup_x")
qi = cols.index("queue_id") if "queue_id" in cols else None
rows.sort(key=lambda r: r[si])
# jobs are separated by gaps > 150 us between a k_copy_out/k_post end and the next k_repack
starts = [i for i, r in enumerate(rows) if "k_repack" in r[ni]]
# last job = the repacks after the last long idle gap
last = starts[-1]
for i in reversed(starts):
    if rows[last][si] - rows[i][si] > 3_000_000: break
    first = i
t0 = rows[first][si]
end = max(r[ei] for r in rows[first:])
print("last job: %d dispatches, span %.3f ms" % (len(rows) - first, (end - t0) / 1e6))
for r in rows[first:]:
    nm = re.sub(r"^void jda::|^jda::", "", r[ni])[:46]
    print("%8.3f %8.3f  %7.1f us  q%s  wgs %6d  %s" % ((r[si] - t0) / 1e6, (r[ei] - t0) / 1e6, (r[ei] - r[si]) / 1e3, r[qi] if qi is not None else "?", r[gx] // max(1, r[wx]), nm))
