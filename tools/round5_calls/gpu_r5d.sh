#!/bin/bash
# round-5 GPU call D: the whole GPU suite, then the measurement set (bench, per-launch dump, rocprofv3 kernel stats serial + overlapped, PMC passes)
mkdir -p gpurun_out/r5d; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 -x > gpurun_out/r5d/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> gpurun_out/r5d/pytest_gpu.log
tail -n 12 gpurun_out/r5d/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5d/smoke.log 2>&1; echo "smoke exit $?"; tail -4 gpurun_out/r5d/smoke.log
bash tools/gpu_final.sh 2>&1 | tail -40
bash tools/gpu_pmc_tcc.sh 2>&1 | tail -25
