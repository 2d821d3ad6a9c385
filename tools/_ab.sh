mkdir -p gpurun_out
Q="--no-cpu-baseline --no-glyph256 --no-forward --no-profile --steps 30 --warmup 5"
run() { n=$1; shift; timeout 200 python bench.py $Q "$@" > gpurun_out/r3_ab_$n.json 2>>gpurun_out/r3_ab.err; python -c "import json,sys; d=json.loads(open('gpurun_out/r3_ab_$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'])"; }
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k batchnorm 2>&1 | tail -5
timeout 300 python tools/bn_probe.py > gpurun_out/r3_bn2.log 2>&1; grep -v 'fast vs' gpurun_out/r3_bn2.log | cut -c1-250
for i in 1 2 3; do
run onepass$i
run twopass$i --knob ln:2=3
done
timeout 300 python -m pytest tests/test_round2_gpu.py -q -x -k "config4 or overlap" 2>&1 | tail -3
