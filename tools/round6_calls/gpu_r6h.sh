#!/bin/bash
# round 6, call h: K9 in evaluation mode (BatchNorm in the conv epilogues) - parity (new test, reference goldens that run eval forwards) + eval forward / glyph256 A/B
R=$(pwd); O=$R/gpurun_out/r6h; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_round6_gpu.py tests/test_round2_gpu.py tests/test_engine_gpu.py tests/test_round3_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 -k "batchnorm or logits or full_size or config4 or golden or oracle or fonts or eval or call_sites or glyph" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -8 $O/pytest.log | cut -c1-300
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-profile"
for i in 1 2; do
  for v in 1 0; do
    $B --knob engine:14=$v > $O/fwd_v${v}_$i.json 2>$O/err_v${v}_$i.log;  python -c "import json;d=json.load(open('$O/fwd_v${v}_$i.json'));g=d.get('glyph256',{});print('fold=$v step', d['ms_per_step'], 'fwd eval', d['forward']['eval']['ms'], 'glyph256', {k:g[k] for k in g if 'ms' in k or 'forward' in k})" | cut -c1-600
  done
done
