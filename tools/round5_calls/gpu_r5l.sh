#!/bin/bash
# round 5, call L: reproducibility of the shipped defaults at full size - forwards (live-row form and dense form) and whole training steps
R=$(pwd); O=$R/gpurun_out/r5l; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
(N=2500 GRAD=1 timeout 600 python tools/repro_probe.py; N=2500 GRAD=0 timeout 600 python tools/repro_probe.py; N=600 timeout 900 python tools/step_repro_probe.py; KNOBS="engine:10=0" N=300 timeout 900 python tools/step_repro_probe.py) > $O/repro.log 2>&1
echo "exit $?"; cut -c1-400 $O/repro.log
