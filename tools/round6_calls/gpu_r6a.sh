#!/bin/bash
# round 6, call a: the phase-A changes (workspace slots, trainer call sites, TN list sizing, stream-K out of the production build) +
# a baseline bench line of this round's first box (with the new `parity` object)
R=$(pwd); O=$R/gpurun_out/r6a; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_round5_gpu.py tests/test_round4_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -15 $O/pytest.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; cut -c1-400 $O/bench.json; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6a/bench.json'))
print('value', d['value'], d['ms_per_step'], 'dense', d.get('dense_rows',{}).get('ms_per_step'))
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'])
print('forward', d['forward']['eval'], d['forward']['train'])
print('parity', json.dumps(d.get('parity'))[:1200])
print('glyph256', {k:v for k,v in d.get('glyph256',{}).items() if k!='kernel_families'})
print({k:(v['ms_per_step'],v['tflops']) for k,v in d['kernel_families'].items()})
PY
