#!/bin/bash
# round 5, call K: stream-K tests + probe log, the stream-K step beside the default step, LayerNorm forward block skip
R=$(pwd); O=$R/gpurun_out/r5k; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -k "stream_k or layernorm_forward" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
timeout 300 python tools/streamk_probe.py all > $O/streamk_probe.log 2>&1; echo "probe exit $?"
B="--steps 12 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256"
for i in 1 2; do
  timeout 300 python bench.py $B > $O/bench_default_$i.json 2> $O/bench_default_$i.err; echo "default $i: $(python -c "import json;d=json.load(open('$O/bench_default_$i.json'));print(d['value'],d['ms_per_step'],d['kernel_families']['gemm_nt']['ms_per_step'],d['kernel_families']['gemm_nt']['tflops'])")"
  timeout 300 python bench.py $B --knob engine:11=1 > $O/bench_streamk_$i.json 2> $O/bench_streamk_$i.err; echo "stream-K $i: $(python -c "import json;d=json.load(open('$O/bench_streamk_$i.json'));print(d['value'],d['ms_per_step'],d['kernel_families']['gemm_nt']['ms_per_step'],d['kernel_families']['gemm_nt']['tflops'])")"
done
