#!/bin/bash
# round 6, call o: the GPU suite alone on the final tree (the log committed as profiles/round6_pytest_gpu.log)
R=$(pwd); O=$R/gpurun_out/r6o; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 2400 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> $O/pytest_gpu.log
tail -n 6 $O/pytest_gpu.log | cut -c1-300
