#!/bin/bash
# vector-memory fill path of the GEMM kernels inside a train step: TA / TCP / TCC busy, stall and latency counters (separate rocprofv3 --pmc passes)
R=$(pwd); mkdir -p $R/gpurun_out/pmcf; cd /tmp; export TMPDIR=/tmp; export PYTHONDONTWRITEBYTECODE=1
B="python $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256 --no-overlap"
i=0
for set in "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_RFIFO_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcf/p$i -o p -- $B > $R/gpurun_out/pmcf/log$i.txt 2>&1
  echo "fill pmc pass $i exit $? $(ls -la $R/gpurun_out/pmcf/p$i/*counter_collection.csv 2>/dev/null | awk '{print $5}')"
  rm -f $R/gpurun_out/pmcf/p$i/p_kernel_trace.csv
  tail -2 $R/gpurun_out/pmcf/log$i.txt | cut -c1-300
done
cd $R; python tools/pmc_fill_summary.py gpurun_out/pmcf | tee gpurun_out/pmcf/fill_counters.md | head -16 | cut -c1-400
find gpurun_out/pmcf -name "*.csv" -size +20M -delete
