#!/bin/bash
# round 6, call c: row-granular live GEMMs (realise_set_engine(10, 2)) - kernel + engine parity, A/B against the block form
R=$(pwd); O=$R/gpurun_out/r6c; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_round6_gpu.py tests/test_round4_gpu.py tests/test_round5_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -12 $O/pytest.log | cut -c1-400
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $O/smoke.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256"
for i in 1 2; do
  $B > $O/ab_rows_$i.json 2>/dev/null;  python -c "import json;d=json.load(open('$O/ab_rows_$i.json'));print('row list  ', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['kernel_families']['gemm_nt']['ms_per_step'])"
  $B --knob engine:10=1 > $O/ab_blk_$i.json 2>/dev/null; python -c "import json;d=json.load(open('$O/ab_blk_$i.json'));print('block list', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['kernel_families']['gemm_nt']['ms_per_step'])"
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256 --dump-launches $O/launches.json > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6c/launches.json'))['gemm_nt']
import collections
c=collections.defaultdict(list)
for x in d: c[round(x['gflop_nominal'],1)].append(x['us'])
for k,v in sorted(c.items()): print(k, len(v), round(sum(v)/len(v),2))
PY
