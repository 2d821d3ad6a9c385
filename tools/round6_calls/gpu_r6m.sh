#!/bin/bash
# round 6, call m: CU pairing (realise_set_nt8p(7, 1)) and pairing + non-temporal B fetches (7, 2) against the plain order, four rounds
R=$(pwd); O=$R/gpurun_out/r6m; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-glyph256"
for i in 1 2 3 4; do
  for v in 0 1 2; do
    $B --knob nt8p:7=$v > $O/ab_v${v}_$i.json 2>$O/err_v${v}_$i.log;  python -c "import json;d=json.load(open('$O/ab_v${v}_$i.json'));print('v=$v', d['ms_per_step'], d['roofline']['avg_launch_us'], d['kernel_families']['gemm_nt']['ms_per_step'], 'fwd eval', d['forward']['eval']['ms'], 'fwd train', d['forward']['train']['ms'])"
  done
done
