#!/bin/bash
# round 6, call j: LDS-free AdamW tile (in-register transpose) - optimizer tests, then A/B: LDS tile / register tile / register tile pipelined
R=$(pwd); O=$R/gpurun_out/r6j; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests/test_round6_gpu.py tests/test_round3_gpu.py tests/test_round4_gpu.py tests/test_round5_gpu.py tests/test_round2_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 -k "pipelined or optimizer or adamw or trainer or call_sites or raw_parameter or trajectory" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -6 $O/pytest.log | cut -c1-300
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256 --no-profile"
for i in 1 2 3; do
  $B --knob ln:6=0 > $O/ab_lds_$i.json 2>$O/err_lds_$i.log;  python -c "import json;d=json.load(open('$O/ab_lds_$i.json'));print('lds tile      ', d['ms_per_step'], d['value'])"
  $B > $O/ab_reg_$i.json 2>$O/err_reg_$i.log;  python -c "import json;d=json.load(open('$O/ab_reg_$i.json'));print('register tile ', d['ms_per_step'], d['value'])"
  $B --knob opt:pipeline=1 > $O/ab_pipe_$i.json 2>$O/err_pipe_$i.log; python -c "import json;d=json.load(open('$O/ab_pipe_$i.json'));print('reg pipelined ', d['ms_per_step'], d['value'])"
done
tail -3 $O/err_pipe_1.log
