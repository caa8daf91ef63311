#!/bin/bash
# ON THE GPU BOX: SQ counters for single conv cases.  usage: bash tools/gpu_pmc_cases.sh <tag> "case|pass" "case|pass" ...
set -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for spec in "$@"; do
  c="${spec%%|*}"; p="${spec##*|}"; i=$((i+1))
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"; do
    rm -rf $out/tmp
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/tmp -o run -- python $GRAFT_REPO_ROOT/tools/bench_conv.py "$c" "$p" > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 3; }
    f=$(find $out/tmp -name "*counter_collection.csv" | head -1)
    echo "=== $c | $p" >> $out/summary.txt
    grep -E "ms .*TFLOP" $out/run.log >> $out/summary.txt
    python $GRAFT_REPO_ROOT/tools/pmc_summary.py "$f" >> $out/summary.txt
  done
done
rm -rf $out/tmp
echo pmc-cases-done
