#!/bin/bash
# round-5 GPU call F: the 8-wave grouped weight-gradient kernel after its waitcnt fix, against the 4-wave one (probe + step A/B)
mkdir -p gpurun_out/r5g; export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r5g
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_round4_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=300 -k "tn or group or conv or wgrad or live_row" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 200 python tools/tn_live_probe.py > $O/tn_live_probe.log 2>&1; cat $O/tn_live_probe.log
B="--steps 12 --warmup 4 --no-cpu-baseline --no-fp32-parity --no-glyph256 --no-forward --no-dense-rows-ab --no-profile"
run() { timeout 200 python bench.py $B $2 > $O/bench_$1.json 2> $O/bench_$1.err; python -c "import json;d=json.load(open('$O/bench_$1.json'));print('$1', d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do run spread_$i ""; done
