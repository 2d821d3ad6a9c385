#!/bin/bash
# round 5, call M: S > 128 - the tiled attention kernels alone and inside the model (reference goldens at S = 256 / 512), live rows at S = 256
R=$(pwd); O=$R/gpurun_out/r5m; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_round5_gpu.py tests/test_round4_gpu.py -q --no-header -rfE -p no:cacheprovider -k "attention or long_sequence or stream_k_training or live_row_training_step" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -25 $O/pytest.log | cut -c1-300
