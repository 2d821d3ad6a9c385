"""Vector-memory fill path per kernel from the rocprofv3 --pmc passes of tools/gpu_pmc_fill.sh: mean per launch of every counter found, plus
L2 read latency seen by the TCPs (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ, cycles), L2 hit rate, and the stall counters as shares of
TCP_GATE_EN1 (cycles the TCPs are clocked)."""
import collections, csv, glob, re, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + '/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*\)$", "", r['Kernel_Name']).replace("rl::", "").replace("void ", "")
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for k, cs in agg.items():
    n = max(len(v) for v in cs.values())
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    rows.append((k, n, m))
rows.sort(key=lambda r: -r[2].get('GRBM_GUI_ACTIVE', 0.0) * r[1])
names = sorted({c for _, _, m in rows for c in m})
print("| kernel | launches | " + " | ".join(names) + " | L2 read latency (clk) | L2 hit rate | TA-data stall / TCP clocked | pending stall / TCP clocked | TCR stall / TCP clocked | TA busy / TCP clocked |")
print("|---|---|" + "---|" * (len(names) + 6))
d = lambda a, b: ("%.3g" % (a / b)) if b else "-"
for k, n, m in rows[:24]:
    g = m.get('TCP_GATE_EN1_sum', 0.0)
    print("| `%s` | %d | " % (k[:80], n) + " | ".join("%.3g" % m.get(c, 0.0) for c in names) + " | %s | %s | %s | %s | %s | %s |" % (
        d(m.get('TCP_TCC_READ_REQ_LATENCY_sum', 0.0), m.get('TCP_TCC_READ_REQ_sum', 0.0)),
        d(m.get('TCC_HIT_sum', 0.0), m.get('TCC_HIT_sum', 0.0) + m.get('TCC_MISS_sum', 0.0)),
        d(m.get('TCP_TCP_TA_DATA_STALL_CYCLES_sum', 0.0), g), d(m.get('TCP_PENDING_STALL_CYCLES_sum', 0.0), g),
        d(m.get('TCP_TCR_TCP_STALL_CYCLES_sum', 0.0), g), d(m.get('TA_TA_BUSY_sum', 0.0), g)))
