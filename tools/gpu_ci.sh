#!/bin/bash
# run on the GPU box through gpurun: kernel + engine parity tests, logs into gpurun_out/
#   bash tools/gpu_ci.sh "<test files>" [timeout_s] ["<-k expression>"]
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
SEL="${1:-tests}"
KEXPR="${3:-}"
if [ -n "$KEXPR" ]; then
  timeout ${2:-1500} python -m pytest $SEL -m gpu -k "$KEXPR" -q --no-header -rfE -p no:cacheprovider --timeout=600 > gpurun_out/pytest_gpu.log 2>&1
else
  timeout ${2:-1500} python -m pytest $SEL -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 > gpurun_out/pytest_gpu.log 2>&1
fi
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -n 120 gpurun_out/pytest_gpu.log
