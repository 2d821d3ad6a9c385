#!/bin/bash
# round 6, call p: L2 prefetch workgroups in the narrow (N = 768) row-list GEMM launches (realise_set_nt8p(8, v): bits 0-3 workgroups per XCD,
# 4-7 K-tiles ahead before pacing, 8-15 pacing sleeps of 256 clocks per K-tile) - bit-identity + per-shape launch times + step
R=$(pwd); O=$R/gpurun_out/r6p; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tools/live_big_check.py 8 > $O/check.log 2>&1; echo "check exit $?"; tail -2 $O/check.log | cut -c1-300
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256"
for v in 0 $((6 + (4<<4))) $((10 + (4<<4))) $((10 + (2<<4) + (2<<8))) $((10 + (4<<4) + (4<<8))) 0 $((10 + (8<<4))); do
  $B --knob nt8p:8=$v --dump-launches $O/launches_$v.json > $O/ab_$v.json 2>$O/err_$v.log
  python - <<PY
import json, collections
d=json.load(open('$O/ab_$v.json')); L=json.load(open('$O/launches_$v.json'))['gemm_nt']
c=collections.defaultdict(list)
for x in L: c[round(x['gflop_nominal'],1)].append(x['us'])
print('v=%5d step %.3f family %.3f |' % ($v, d['ms_per_step'], d['kernel_families']['gemm_nt']['ms_per_step']), ' '.join('%s GF: %d x %.1f us' % (k, len(u), sum(u)/len(u)) for k,u in sorted(c.items()) if k < 100))
PY
done
