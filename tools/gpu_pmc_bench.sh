#!/bin/bash
# ON THE GPU BOX: SQ counters (two passes) for every kernel of one bench.py workload.
#   usage: bash tools/gpu_pmc_bench.sh <tag> <bench.py args...>      -> gpurun_out/<tag>_sq.txt
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $out/${tag}_sq.txt
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"; do
  rm -rf $out/${tag}_tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/${tag}_tmp -o run -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $out/${tag}_pmc.log 2>&1 || { tail -5 $out/${tag}_pmc.log; exit 3; }
  f=$(find $out/${tag}_tmp -name "*counter_collection.csv" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py "$f" >> $out/${tag}_sq.txt
done
rm -rf $out/${tag}_tmp
echo pmc-bench-done
