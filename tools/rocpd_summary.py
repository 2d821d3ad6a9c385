"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration.
usage: python tools/rocpd_summary.py <results.db> [steps] > profiles/<name>.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)


def short(n):
    n = re.sub(r"\(.*\)$", "", n)
    n = n.replace("rl::", "").replace("void ", "")
    return n[:110]


print("| kernel | calls | calls/step | total ms | ms/step | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|---|---|")
for name, calls, tot, avg, mn, mx in rows:
    print("| `%s` | %d | %.1f | %.3f | %.3f | %.2f | %.2f | %.2f | %.1f |" % (short(name), calls, calls / steps, tot / 1e6, tot / 1e6 / steps, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
print("\ntotal kernel time: %.3f ms (%.3f ms/step over %g steps incl. warm-up)" % (total / 1e6, total / 1e6 / steps, steps))
