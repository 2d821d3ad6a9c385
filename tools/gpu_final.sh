#!/bin/bash
# end-of-round measurement set (one gpurun call): default bench, glyph256 bench, rocprofv3 kernel stats (serial + overlapped), PMC passes
R=$(pwd); mkdir -p $R/gpurun_out/final; export PYTHONDONTWRITEBYTECODE=1
python bench.py > $R/gpurun_out/final/bench.json 2> $R/gpurun_out/final/bench.err; echo "bench exit $?"; cut -c1-400 $R/gpurun_out/final/bench.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256 --dump-launches $R/gpurun_out/final/launches.json > /dev/null 2>&1
bash tools/gpu_rocprof.sh fs --no-overlap --no-forward --no-glyph256 > /dev/null 2>&1; cp gpurun_out/prof_fs/kernel_summary.md gpurun_out/final/kernel_stats_serial.md
bash tools/gpu_rocprof.sh fo --no-forward --no-glyph256 > /dev/null 2>&1; cp gpurun_out/prof_fo/kernel_summary.md gpurun_out/final/kernel_stats_overlap.md
tail -1 gpurun_out/final/kernel_stats_serial.md; tail -1 gpurun_out/final/kernel_stats_overlap.md
rm -rf gpurun_out/prof_fs gpurun_out/prof_fo
bash tools/gpu_pmc_bench.sh > gpurun_out/final/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmcb > gpurun_out/final/pmc_counters.md 2> gpurun_out/final/pmc_summary.err; cp gpurun_out/pmcb/traffic.json gpurun_out/final/pmc_traffic.json
head -12 gpurun_out/final/pmc_counters.md | cut -c1-220
rm -rf gpurun_out/pmcb/p1 gpurun_out/pmcb/p2 gpurun_out/pmcb/p3
