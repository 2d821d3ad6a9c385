#!/bin/bash
# round-5 GPU call B: new tests + regression subset, LayerNorm probe, step A/B of this round's knobs
mkdir -p gpurun_out/r5c; export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r5c
timeout 600 python -m pytest tests/test_round5_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=300 > $O/pytest_new.log 2>&1; echo "new tests exit $?" >> $O/pytest_new.log
tail -n 25 $O/pytest_new.log
timeout 700 python -m pytest tests/test_kernels_gpu.py tests/test_round4_gpu.py tests/test_engine_gpu.py tests/test_round3_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=300 \
  -k "layernorm or padding or live_row or grouped or trajectory or fused_adamw or full_size or reference_trainer or gemm_nt or nt8 or split_k" > $O/pytest_reg.log 2>&1; echo "regression exit $?" >> $O/pytest_reg.log
tail -n 8 $O/pytest_reg.log
timeout 120 python tools/ln_probe.py > $O/ln_probe.log 2>&1; cat $O/ln_probe.log
B="--steps 12 --warmup 4 --no-cpu-baseline --no-fp32-parity --no-glyph256 --no-forward --no-dense-rows-ab --no-profile"
run() { timeout 200 python bench.py $B $2 > $O/bench_$1.json 2> $O/bench_$1.err; python -c "import json;d=json.load(open('$O/bench_$1.json'));print('$1', d['ms_per_step'], d['value'])"; }
for i in 1 2; do
  run default_$i ""
  run ln_v1_$i "--knob ln:5=0"
  run bias_item_$i "--knob nt8p:4=0"
  run xcd1d_$i "--knob nt8p:3=1"
done
