#!/bin/bash
# round 6, call i: pipelined optimizer sweep - bit-identity test, optimizer / trainer tests, A/B of the step
R=$(pwd); O=$R/gpurun_out/r6i; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests/test_round6_gpu.py tests/test_round3_gpu.py tests/test_round4_gpu.py tests/test_round5_gpu.py tests/test_round2_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 -k "pipelined or optimizer or adamw or trainer or call_sites or raw_parameter or trajectory or logits" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -6 $O/pytest.log | cut -c1-300
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256 --no-profile"
for i in 1 2 3; do
  $B > $O/ab_pipe_$i.json 2>$O/err_pipe_$i.log;  python -c "import json;d=json.load(open('$O/ab_pipe_$i.json'));print('pipelined', d['ms_per_step'], d['value'])"
  $B --knob opt:pipeline=0 > $O/ab_plain_$i.json 2>$O/err_plain_$i.log; python -c "import json;d=json.load(open('$O/ab_plain_$i.json'));print('plain    ', d['ms_per_step'], d['value'])"
done
tail -3 $O/err_pipe_1.log
