#!/bin/bash
# rocprofv3 kernel trace + stats of the bench (no HIP-event bracketing, serial branches so every kernel is timed alone)
#   bash tools/gpu_rocprof.sh <tag> [extra bench args]
R=$(pwd); TAG=${1:-r2}; shift
mkdir -p $R/gpurun_out/prof_$TAG
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-fp32-parity --no-dense-rows-ab "$@" > $R/gpurun_out/prof_$TAG/bench.json 2> $R/gpurun_out/prof_$TAG/bench.err
echo "rocprof exit $?"; cat $R/gpurun_out/prof_$TAG/bench.json | cut -c1-300
cd $R
db=$(find gpurun_out/prof_$TAG -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_summary.py $db 7 > gpurun_out/prof_$TAG/kernel_summary.md; head -45 gpurun_out/prof_$TAG/kernel_summary.md; tail -2 gpurun_out/prof_$TAG/kernel_summary.md; fi
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/prof_$TAG/kernel_stats.csv
find gpurun_out/prof_$TAG -name "*.db" -size +40M -delete
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
