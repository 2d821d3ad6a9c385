#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5o; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_engine_gpu.py -q --no-header -rfE -p no:cacheprovider -k "layernorm_forward_skips or rejects_sequences" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -12 $O/pytest.log | cut -c1-250
