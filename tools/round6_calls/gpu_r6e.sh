#!/bin/bash
# round 6, call e: K4 (GEMM + residual + LayerNorm in one launch) with the L2 invalidate between polls - dense-row step and eval forward A/B
R=$(pwd); O=$R/gpurun_out/r6e; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_round4_gpu.py tests/test_round6_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 -k "layernorm or logits" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log | cut -c1-300
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-glyph256 --no-profile --dense-rows"
for i in 1 2; do
  $B > $O/ab_k4off_$i.json 2>$O/err_off_$i.log;  python -c "import json;d=json.load(open('$O/ab_k4off_$i.json'));print('K4 off', d['ms_per_step'], d.get('forward'))" | cut -c1-400
  $B --knob engine:8=1 > $O/ab_k4on_$i.json 2>$O/err_on_$i.log; python -c "import json;d=json.load(open('$O/ab_k4on_$i.json'));print('K4 on ', d['ms_per_step'], d.get('forward'))" | cut -c1-400
done
tail -3 $O/err_on_1.log
