#!/bin/bash
# kernel trace of an OVERLAPPED run (three branch streams + weight-gradient side stream) and its per-queue timeline
R=$(pwd); TAG=${1:-tl}; shift
mkdir -p $R/gpurun_out/prof_$TAG
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-forward --no-glyph256 "$@" > $R/gpurun_out/prof_$TAG/bench.json 2> $R/gpurun_out/prof_$TAG/bench.err
cd $R
db=$(find gpurun_out/prof_$TAG -name "*.db" | head -1)
python tools/timeline.py $db 1 > gpurun_out/prof_$TAG/timeline.txt 2>&1
python tools/rocpd_summary.py $db 6 > gpurun_out/prof_$TAG/kernel_summary.md
find gpurun_out/prof_$TAG -name "*.db" -delete
cat gpurun_out/prof_$TAG/timeline.txt
