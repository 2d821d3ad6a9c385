#!/bin/bash
# round 6, call d: K13 (training forward without [B, S, V] logits) - parity tests + A/B of the step; K7 test under the run-to-run check
R=$(pwd); O=$R/gpurun_out/r6d; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_round6_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 -k "logits or glyph_lookup" > $O/pytest6.log 2>&1; echo "pytest6 exit $?"; tail -15 $O/pytest6.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $O/smoke.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256 --no-profile"
for i in 1 2; do
  $B > $O/ab_k13on_$i.json 2>$O/err_on_$i.log;  python -c "import json;d=json.load(open('$O/ab_k13on_$i.json'));print('K13 on ', d['ms_per_step'])"
  $B --knob opt:train_logits=1 > $O/ab_k13off_$i.json 2>$O/err_off_$i.log; python -c "import json;d=json.load(open('$O/ab_k13off_$i.json'));print('K13 off', d['ms_per_step'])"
done
tail -3 $O/err_on_1.log
