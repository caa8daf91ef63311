#!/bin/bash
# GPU box: time the fp16 conv kernels with parts switched off (MSTG_F16_DBG bits: 1 no K-steps, 2 no stores, 4 no fetch, 8 no commit)
cd $GRAFT_REPO_ROOT
for m in 0 1 7 15; do
  echo "=== MSTG_F16_DBG=$m"
  MSTG_F16_DBG=$m python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --kernel-table 2>&1 | grep -E "^\[kernels\] conv_f16" | awk '{printf "%s %s %s %s %s | %s ms %s GB/s\n", $2,$3,$4,$5,$6,$(NF-6),$(NF-1)}'
done
