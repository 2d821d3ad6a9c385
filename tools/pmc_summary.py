"""Aggregate rocprofv3 --pmc counter_collection csv files per kernel: mean per launch.
usage: python tools/pmc_summary.py gpurun_out/pmcb > profiles/<name>.md ; also writes <dir>/traffic.json"""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernels_sha      # the summary is stamped with the hash of the kernel sources it was measured on
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
    # MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte requests at 64 B -> x2
    rd = m.get('FETCH_SIZE', 0.0) * 1024 * 2
    wr = m.get('WRITE_SIZE', 0.0) * 1024
    rows.append((k, n, rd, wr, m))
rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
print("| kernel | launches | HBM read MB/launch (FETCH_SIZE x2) | HBM write MB/launch | MFMA busy % of wave cycles | wait_any % | wait_inst % | active % | LDS bank conflict |")
print("|---|---|---|---|---|---|---|---|---|")
out = {}
for k, n, rd, wr, m in rows[:40]:
    wc = m.get('SQ_WAVE_CYCLES', 0.0) * 4.0
    mf = 100.0 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / wc if wc else 0.0
    f = lambda c: (100.0 * m.get(c, 0.0) / m['SQ_WAVE_CYCLES']) if m.get('SQ_WAVE_CYCLES') else 0.0
    print("| `%s` | %d | %.2f | %.2f | %.1f | %.1f | %.1f | %.1f | %.0f |" % (k[:90], n, rd / 1e6, wr / 1e6, mf, f('SQ_WAIT_ANY'), f('SQ_WAIT_INST_ANY'), f('SQ_ACTIVE_INST_ANY'), m.get('SQ_LDS_BANK_CONFLICT', 0.0)))
    out[k] = {"launches": n, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr}
json.dump({"kernels_sha": kernels_sha(), "command": "rocprofv3 --pmc <set> --kernel-trace -- python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-forward --no-glyph256 (one pass per counter set)",
           "units": "bytes per launch; FETCH_SIZE KiB x 2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE KiB", "kernels": out}, open(root + '/traffic.json', 'w'), indent=1)
