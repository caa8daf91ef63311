#!/bin/bash
# GPU box: A/B of two builds of the library in ONE call (numbers across boxes differ by ~10 %).
#   usage: bash tools/f16_ab.sh [bench args]      compares mstg_hip/libmstg_hip_prev.so (A) with the current build (B)
cd $GRAFT_REPO_ROOT
prev=$GRAFT_REPO_ROOT/multi-style-transfer-gan_amd/mstg_hip/libmstg_hip_prev.so
for v in A B A B; do
  echo "=== build $v"
  if [ $v = A ]; then export MSTG_LIB=$prev; else unset MSTG_LIB; fi
  if [ $# -eq 0 ]; then set -- --config 5; fi
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --kernel-table 2>&1 | grep -E "^\[kernels\]|\"value\"" | awk '{ if ($1=="[kernels]") printf "%s %s %s %s %s | %s ms\n", $2,$3,$4,$5,$6,$(NF-6); else print substr($0, 1, 140) }'
done
