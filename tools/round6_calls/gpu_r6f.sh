#!/bin/bash
# round 6, call f: wide row-list GEMMs on 256 x 256 one-per-CU tiles (realise_set_nt8p(5, v)): correctness + A/B of the step
R=$(pwd); O=$R/gpurun_out/r6f; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tools/live_big_check.py > $O/check.log 2>&1; echo "check exit $?"; tail -3 $O/check.log | cut -c1-300
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256"
for i in 1 2; do
  for v in 0 1 2; do
    $B --knob nt8p:5=$v > $O/ab_v${v}_$i.json 2>$O/err_v${v}_$i.log; python -c "import json;d=json.load(open('$O/ab_v${v}_$i.json'));print('v=$v', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['kernel_families']['gemm_nt']['ms_per_step'])"
  done
done
