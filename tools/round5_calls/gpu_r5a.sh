#!/bin/bash
# round-5 GPU call A: the new tests, the LayerNorm / live-row regression tests, the LayerNorm probe, a short A/B of the step
mkdir -p gpurun_out/r5a; export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r5a
timeout 600 python -m pytest tests/test_round5_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=300 > $O/pytest_new.log 2>&1; echo "new tests exit $?" >> $O/pytest_new.log
tail -n 40 $O/pytest_new.log
timeout 500 python -m pytest tests/test_kernels_gpu.py tests/test_round4_gpu.py tests/test_engine_gpu.py tests/test_round3_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=300 \
  -k "layernorm or padding or live_row or grouped or trajectory or fused_adamw or full_size or reference_trainer" > $O/pytest_reg.log 2>&1; echo "regression exit $?" >> $O/pytest_reg.log
tail -n 15 $O/pytest_reg.log
timeout 120 python tools/ln_probe.py > $O/ln_probe.log 2>&1; cat $O/ln_probe.log
B="--steps 12 --warmup 4 --no-cpu-baseline --no-fp32-parity --no-glyph256 --no-forward --no-dense-rows-ab --no-profile"
for i in 1 2; do
  timeout 200 python bench.py $B > $O/bench_v2_$i.json 2> $O/bench_v2_$i.err; python -c "import json;d=json.load(open('$O/bench_v2_$i.json'));print('v2 ', d['ms_per_step'], d['value'])"
  timeout 200 python bench.py $B --knob ln:5=0 > $O/bench_v1_$i.json 2> $O/bench_v1_$i.err; python -c "import json;d=json.load(open('$O/bench_v1_$i.json'));print('v1 ', d['ms_per_step'], d['value'])"
done
for i in 1 2; do
  timeout 200 python bench.py $B --knob nt8p:3=1 > $O/bench_gc1_$i.json 2> $O/bench_gc1_$i.err; python -c "import json;d=json.load(open('$O/bench_gc1_$i.json'));print('xcd 1-D ', d['ms_per_step'], d['value'])"
  timeout 200 python bench.py $B > $O/bench_gc0_$i.json 2> $O/bench_gc0_$i.err; python -c "import json;d=json.load(open('$O/bench_gc0_$i.json'));print('xcd auto', d['ms_per_step'], d['value'])"
done
