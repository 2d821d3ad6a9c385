#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5r; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_round2_gpu.py tests/test_engine_gpu.py -q --no-header -rfE -p no:cacheprovider -k "pinyin_width or trainer_call_sites or host_batch or pho" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -12 $O/pytest.log | cut -c1-250
