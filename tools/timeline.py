"""Per-queue timeline of ONE training step from a rocprofv3 rocpd database (kernel trace of an overlapped run): which HIP stream
(hardware queue) is busy when, and for how long the caller's stream waits with nothing to do.
    python tools/timeline.py <results.db> [step_index_from_end]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = db.execute("select name, queue_id, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if r[0].startswith("rl::adamw") or "adamw_kernel" in r[0]]
if len(marks) < back + 1:
    raise SystemExit("not enough steps in the trace")
lo, hi = marks[-back - 1] + 1, marks[-back] + 1
step = rows[lo:hi]
t0 = step[0][2]
T = (step[-1][3] - t0) / 1e3
print("step: %d kernels, %.1f us wall" % (len(step), T))
queues = sorted({r[1] for r in step})


def short(n):
    n = re.sub(r"\(.*\)$", "", n).replace("rl::", "").replace("void ", "")
    return re.sub(r"<.*", "", n)[:28]


for q in queues:
    ks = [r for r in step if r[1] == q]
    busy = sum(r[3] - r[2] for r in ks) / 1e3
    print("queue %d: %4d kernels, busy %8.1f us, first start %8.1f, last end %8.1f" % (q, len(ks), busy, (ks[0][2] - t0) / 1e3, (ks[-1][3] - t0) / 1e3))
# windows of 250 us: busy fraction per queue and the dominant kernel of the busiest queue
W = 250.0
nw = int(T / W) + 1
print("\nwindow(us)  " + "  ".join("q%-5d" % q for q in queues) + "  dominant")
for w in range(nw):
    a, b = t0 + w * W * 1e3, t0 + (w + 1) * W * 1e3
    cells, dom = [], {}
    for q in queues:
        t = 0.0
        for r in step:
            if r[1] != q:
                continue
            o = min(r[3], b) - max(r[2], a)
            if o > 0:
                t += o
                dom[short(r[0])] = dom.get(short(r[0]), 0.0) + o
        cells.append("%5.0f%%" % (100.0 * t / (W * 1e3)))
    top = sorted(dom.items(), key=lambda kv: -kv[1])[:3]
    print("%6.0f      %s  %s" % (w * W, "  ".join(cells), ", ".join("%s %.0f" % (k, v / 1e3) for k, v in top)))
# idle gaps of the whole device (no queue busy)
ev = sorted([(r[2], 1) for r in step] + [(r[3], -1) for r in step])
depth, last, idle = 0, t0, 0.0
for t, d in ev:
    if depth == 0 and t > last:
        idle += t - last
    depth += d
    last = t
print("\ndevice idle inside the step: %.1f us" % (idle / 1e3))
