#!/bin/bash
# round 6, call k: grouped weight gradients walking the wide-J problem j-panel by j-panel (realise_set_nt8p(6, v)) - tests + A/B
R=$(pwd); O=$R/gpurun_out/r6k; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_round4_gpu.py tests/test_round5_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 -k "grouped or group or weight_grad or live_row_training_step" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log | cut -c1-300
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256"
for i in 1 2 3; do
  $B > $O/ab_j1_$i.json 2>$O/err_j1_$i.log;  python -c "import json;d=json.load(open('$O/ab_j1_$i.json'));f=d['kernel_families']['gemm_tn'];print('j-major', d['ms_per_step'], f['ms_per_step'], f.get('avg_launch_us'))"
  $B --knob nt8p:6=0 > $O/ab_j0_$i.json 2>$O/err_j0_$i.log; python -c "import json;d=json.load(open('$O/ab_j0_$i.json'));f=d['kernel_families']['gemm_tn'];print('i-major', d['ms_per_step'], f['ms_per_step'], f.get('avg_launch_us'))"
done
