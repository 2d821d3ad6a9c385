#!/bin/bash
# round 5, call Q: the fused dense + LayerNorm launch (K4, form 2: off by default) under the reproducibility probe at scale
R=$(pwd); O=$R/gpurun_out/r5q; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
(KNOBS="engine:8=1 engine:10=0" N=2500 GRAD=0 timeout 600 python tools/repro_probe.py; KNOBS="engine:8=1 engine:10=0" N=300 timeout 600 python tools/step_repro_probe.py) > $O/repro_k4.log 2>&1
echo "exit $?"; grep -v amdgpu.ids $O/repro_k4.log | cut -c1-400
