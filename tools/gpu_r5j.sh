#!/bin/bash
R=$(pwd); mkdir -p $R/gpurun_out/r5j; export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tools/_sk_dbg2.py > $R/gpurun_out/r5j/dbg.log 2>&1; echo "exit $?"; cut -c1-400 $R/gpurun_out/r5j/dbg.log
