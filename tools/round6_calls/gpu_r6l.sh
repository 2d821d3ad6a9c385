#!/bin/bash
# round 6, call l: CU pairing of column tiles (realise_set_nt8p(7, 1)) - bit-identity check + A/B of the step and the forward
R=$(pwd); O=$R/gpurun_out/r6l; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tools/live_big_check.py 7 > $O/check.log 2>&1; echo "check exit $?"; tail -2 $O/check.log | cut -c1-300
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-glyph256"
for i in 1 2 3 4; do
  $B > $O/ab_p0_$i.json 2>$O/err_p0_$i.log;  python -c "import json;d=json.load(open('$O/ab_p0_$i.json'));print('plain ', d['ms_per_step'], d['roofline']['avg_launch_us'], d['kernel_families']['gemm_nt']['ms_per_step'], 'fwd eval', d['forward']['eval']['ms'])"
  $B XX, d['ms_per_step'], d['roofline']['avg_launch_us'], d['kernel_families']['gemm_nt']['ms_per_step'], 'fwd eval', d['forward']['eval']['ms'])"
done
