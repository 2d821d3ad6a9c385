#!/bin/bash
# round 5, call N: the stream-K step test once more, and the training step at S = 256 / 512 (same 8192 tokens per step as the headline batch)
R=$(pwd); O=$R/gpurun_out/r5n; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests/test_round5_gpu.py -q --no-header -p no:cacheprovider -k "stream_k_training" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log | cut -c1-200
B="--steps 10 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256"
for cfg in "64 128" "32 256" "16 512"; do set -- $cfg
  timeout 300 python bench.py $B --batch $1 --seq $2 > $O/bench_b$1_s$2.json 2> $O/bench_b$1_s$2.err; echo "B $1 S $2: $(python -c "import json;d=json.load(open('$O/bench_b$1_s$2.json'));k=d['kernel_families'];print(d['value'],'sentences/s',d['ms_per_step'],'ms/step; attention fwd',k['attn_fwd']['ms_per_step'],'bwd',k['attn_bwd']['ms_per_step'],'ms; mean loss',d['config'].get('mean_loss'))" 2>&1 | tail -1)"
done
