#!/bin/bash
# end-of-round set, one gpurun call: the GPU suite, smoke, then the measurements in the order that lets the committed bench line carry them:
# PMC passes -> profiles/round6_pmc_traffic.json (on the box), rocprofv3 kernel stats -> profiles/round6_final_kernel_stats.md (on the box),
# THEN python bench.py (roofline.traffic / roofline.rocprof resolve), per-launch dump.  Everything is copied to gpurun_out/final6.
# (The L2 / TCC passes of tools/gpu_pmc_tcc.sh do not depend on the kernel-source hash: profiles/round6_pmc_tcc.md is from the run before.)
R=$(pwd); O=$R/gpurun_out/final6; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 1800 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider --timeout=600 > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> $O/pytest_gpu.log
tail -n 6 $O/pytest_gpu.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $O/smoke.log
bash tools/gpu_pmc_bench.sh > $O/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmcb > $O/pmc_counters.md 2> $O/pmc_summary.err; cp gpurun_out/pmcb/traffic.json $O/pmc_traffic.json; cp gpurun_out/pmcb/traffic.json profiles/round6_pmc_traffic.json
rm -rf gpurun_out/pmcb/p1 gpurun_out/pmcb/p2 gpurun_out/pmcb/p3
bash tools/gpu_rocprof.sh fs --no-overlap --no-forward --no-glyph256 > /dev/null 2>&1; cp gpurun_out/prof_fs/kernel_summary.md $O/kernel_stats_serial.md; cp gpurun_out/prof_fs/kernel_summary.md profiles/round6_final_kernel_stats.md
bash tools/gpu_rocprof.sh fo --no-forward --no-glyph256 > /dev/null 2>&1; cp gpurun_out/prof_fo/kernel_summary.md $O/kernel_stats_overlap.md
tail -1 $O/kernel_stats_serial.md; tail -1 $O/kernel_stats_overlap.md
rm -rf gpurun_out/prof_fs gpurun_out/prof_fo
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; cut -c1-300 $O/bench.json; cp $O/bench.json profiles/round6_final_bench.json
python -c "import json;d=json.load(open('$O/bench.json'));r=d['roofline'];print('roofline', r['frac'], r['achieved'], r['traffic'], r.get('traffic_source'), r.get('rocprof'))"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256 --dump-launches $O/launches.json > /dev/null 2>&1
head -6 $O/pmc_counters.md | cut -c1-220
