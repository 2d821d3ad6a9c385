#!/bin/bash
# the GPU suite alone on the final tree (the measurement set of tools/gpu_final3.sh is from the same kernel sources)
R=$(pwd); O=$R/gpurun_out/r5p; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> $O/pytest_gpu.log
tail -n 8 $O/pytest_gpu.log | cut -c1-300
