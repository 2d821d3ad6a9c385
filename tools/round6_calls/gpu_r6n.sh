#!/bin/bash
# round 6, call n: live-row evaluation forward (model.eval_live_rows) - tests + forward block of the bench
R=$(pwd); O=$R/gpurun_out/r6n; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_round2_gpu.py tests/test_engine_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 -k "evaluation or eval or logits or full_size or golden or oracle or band or taps or call_sites" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log | cut -c1-300
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-glyph256 --no-profile"
for i in 1 2; do
  $B > $O/fwd_$i.json 2>$O/err_$i.log;  python -c "import json;d=json.load(open('$O/fwd_$i.json'));print('step', d['ms_per_step'], {k: (v['ms'], v['mfma_util_nominal'], v['mfma_util_executed']) for k, v in d['forward'].items() if isinstance(v, dict)})"
done
