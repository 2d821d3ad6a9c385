#!/bin/bash
# round 5, call I: stream-K layer GEMM probe (results + time per launch against the 128 x 192 kernels)
R=$(pwd); mkdir -p $R/gpurun_out/r5i; export PYTHONDONTWRITEBYTECODE=1
timeout 420 python tools/streamk_probe.py all > $R/gpurun_out/r5i/streamk_probe.log 2>&1; echo "probe exit $?"
cat $R/gpurun_out/r5i/streamk_probe.log | cut -c1-330
