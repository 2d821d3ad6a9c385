#!/bin/bash
# round 6, call b: K7 (glyph lookup fused into block 1's forward conv loaders) - parity tests that touch the glyph branch + an A/B of the step
R=$(pwd); O=$R/gpurun_out/r6b; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_round6_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 -s > $O/pytest6.log 2>&1; echo "pytest6 exit $?"; grep -n "trainer lines\|passed\|failed" $O/pytest6.log | cut -c1-900
timeout 1200 python -m pytest tests/test_round2_gpu.py tests/test_round3_gpu.py tests/test_engine_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 -k "glyph or resnet or config2 or config4 or full_size or reference_golden or oracle or fonts" > $O/pytest_glyph.log 2>&1; echo "pytest glyph exit $?"; tail -8 $O/pytest_glyph.log | cut -c1-300
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256 --no-profile"
for i in 1 2; do
  $B > $O/ab_k7on_$i.json 2>/dev/null;  python -c "import json;d=json.load(open('$O/ab_k7on_$i.json'));print('K7 on ', d['ms_per_step'])"
  $B --knob engine:13=0 > $O/ab_k7off_$i.json 2>/dev/null; python -c "import json;d=json.load(open('$O/ab_k7off_$i.json'));print('K7 off', d['ms_per_step'])"
done
