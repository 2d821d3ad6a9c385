#!/bin/bash
# round-5 GPU call E: the TN (weight-gradient) kernels after the waitcnt fixes - parity subset, per-kernel probes, step time
mkdir -p gpurun_out/r5e; export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r5e
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_ops_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py tests/test_round4_gpu.py tests/test_engine_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=400 \
  -k "tn or conv or wgrad or group or resnet or glyph or full_size or padding or live_row or trajectory or gradient or train" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 12 $O/pytest.log
timeout 200 python tools/tn_live_probe.py > $O/tn_live_probe.log 2>&1; head -12 $O/tn_live_probe.log
timeout 200 python tools/conv_tn_probe.py > $O/conv_tn_probe.log 2>&1; tail -25 $O/conv_tn_probe.log
B="--steps 12 --warmup 4 --no-cpu-baseline --no-fp32-parity --no-glyph256 --no-forward --no-dense-rows-ab"
for i in 1 2; do timeout 200 python bench.py $B > $O/bench_$i.json 2> $O/bench_$i.err; python -c "
import json;d=json.load(open('$O/bench_$i.json'));print('step', d['ms_per_step'], d['value']); f=d['kernel_families']; print({k:(round(v.get('tflops',0),1), round(v.get('ms_per_step',0),3)) for k,v in f.items()})"; done
