"""LDS-array occupancy per kernel from one rocprofv3 --pmc pass (tools/gpu_pmc_lds.sh): mean per launch.
SQ_LDS_IDX_ACTIVE = LDS-array cycles (MI355X_MICROARCH.md, LDS section), SQ_LDS_BANK_CONFLICT = the extra cycles among them,
SQ_BUSY_CU_CYCLES = cycles a CU has a wave (both per CU, summed over the CUs), SQ_VALU_MFMA_BUSY_CYCLES per SIMD (x 4 per CU)."""
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
rows.sort(key=lambda r: -r[2].get('SQ_LDS_IDX_ACTIVE', 0.0) * r[1])
names = sorted({c for _, _, m in rows for c in m})
print("| kernel | launches | " + " | ".join(names) + " | LDS cycles / CU-busy cycles | conflict share | LDS cycles / MFMA-busy cycles (per SIMD) |")
print("|---|---|" + "---|" * (len(names) + 3))
for k, n, m in rows[:24]:
    idx, cu = m.get('SQ_LDS_IDX_ACTIVE', 0.0), m.get('SQ_BUSY_CU_CYCLES', 0.0)
    mf = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    print("| `%s` | %d | " % (k[:80], n) + " | ".join("%.3g" % m.get(c, 0.0) for c in names) +
          " | %s | %s | %s |" % ("%.3f" % (idx / cu) if cu else "-", "%.3f" % (m.get('SQ_LDS_BANK_CONFLICT', 0.0) / idx) if idx else "-", "%.2f" % (idx / mf) if mf else "-"))
