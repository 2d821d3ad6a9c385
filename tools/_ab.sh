mkdir -p gpurun_out
Q="--no-cpu-baseline --no-glyph256 --no-forward --no-profile --steps 30 --warmup 5"
run() { n=$1; shift; timeout 200 python bench.py $Q "$@" > gpurun_out/r3_ab_$n.json 2>>gpurun_out/r3_ab.err; python -c "import json,sys; d=json.loads(open('gpurun_out/r3_ab_$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'])"; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "dropout or attention or layernorm" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_engine_gpu.py -q -x -k "dropout" 2>&1 | tail -3
for i in 1 2 3; do run hash4_$i; done
