#!/bin/bash
# PMC passes over a short bench run (separate passes, kernel-trace + csv only), then the final bench + kernel stats
R=$(pwd); mkdir -p $R/gpurun_out/pmcb; cd /tmp; export TMPDIR=/tmp; export PYTHONDONTWRITEBYTECODE=1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcb/p$i -o p -- python $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256 > $R/gpurun_out/pmcb/log$i.txt 2>&1
  echo "pmc pass $i exit $?"
  rm -f $R/gpurun_out/pmcb/p$i/p_kernel_trace.csv
done
du -sh $R/gpurun_out/pmcb
