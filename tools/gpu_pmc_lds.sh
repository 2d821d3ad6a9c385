#!/bin/bash
# LDS-array occupancy of the GEMM kernels inside a train step (one rocprofv3 --pmc pass; a second, smaller set if a counter name is unknown)
R=$(pwd); mkdir -p $R/gpurun_out/pmcl; cd /tmp; export TMPDIR=/tmp; export PYTHONDONTWRITEBYTECODE=1
B="python $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-fp32-parity --no-dense-rows-ab --no-forward --no-glyph256 --no-overlap"
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcl/p$i -o p -- $B > $R/gpurun_out/pmcl/log$i.txt 2>&1
  echo "lds pmc pass $i exit $?"
  rm -f $R/gpurun_out/pmcl/p$i/p_kernel_trace.csv
  if [ -s $R/gpurun_out/pmcl/p$i/p_counter_collection.csv ]; then break; fi
done
cd $R; python tools/pmc_lds_summary.py gpurun_out/pmcl | tee gpurun_out/pmcl/lds_counters.md | head -24
find gpurun_out/pmcl -name "*.csv" -size +20M -delete
