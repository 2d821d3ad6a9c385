#!/bin/bash
# bench + rocprofv3 kernel stats on the GPU box (through gpurun); results under gpurun_out/
R=$(pwd)
mkdir -p $R/gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python bench.py --gpus 1 --steps ${1:-10} --warmup 3 > $R/gpurun_out/bench.json 2> $R/gpurun_out/bench.err
echo "bench exit $?"; cat $R/gpurun_out/bench.json; tail -5 $R/gpurun_out/bench.err
if [ "${2:-prof}" = "prof" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-fp32-parity --no-dense-rows-ab > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
  echo "rocprof exit $?"; cat $R/gpurun_out/prof_bench.json
  find $R/gpurun_out/prof -name "*stats*" | head; 
  f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1); head -40 "$f"
  # keep only the small summaries
  find $R/gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
