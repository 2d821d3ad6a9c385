#!/bin/bash
# the first 8-GPU run: one table of bench.py over gpus {1,2,4,8} x --ddp-algo {allreduce,mesh} x --grad-dtype {fp32,bf16} with exposed_tail_ms
exec python "$(dirname "$0")/scale_sweep.py" "$@"
