#!/bin/bash
# L2 (TCC) counters of the GEMM kernels inside a train step - VERDICT round 4 item 3.  Round 4's single TCC pass asked for five TCC
# counters (the block has FOUR slots, MI355X_MICROARCH.md) and hung; here every pass carries at most three, under its own timeout.
R=$(pwd); mkdir -p $R/gpurun_out/pmct; cd /tmp; export TMPDIR=/tmp; export PYTHONDONTWRITEBYTECODE=1
B="python $R/bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256 --no-overlap"
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u > $R/gpurun_out/pmct/tcc_counters_available.txt
i=0
for set in "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmct/p$i -o p -- $B > $R/gpurun_out/pmct/log$i.txt 2>&1
  echo "tcc pmc pass $i ($set) exit $? $(ls -la $R/gpurun_out/pmct/p$i/*counter_collection.csv 2>/dev/null | awk '{print $5}')"
  rm -f $R/gpurun_out/pmct/p$i/p_kernel_trace.csv
  tail -2 $R/gpurun_out/pmct/log$i.txt | cut -c1-200
done
cd $R; python tools/pmc_fill_summary.py gpurun_out/pmct > gpurun_out/pmct/tcc_counters.md 2> gpurun_out/pmct/summary.err; head -12 gpurun_out/pmct/tcc_counters.md | cut -c1-420
find gpurun_out/pmct -name "*.csv" -size +20M -delete
