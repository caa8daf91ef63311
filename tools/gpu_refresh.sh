#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench with per-kernel table, rocprofv3 kernel stats, two PMC passes.
# usage: bash tools/gpu_refresh.sh <tag>
set -o pipefail
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-table > $out/${tag}_bench.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_prof $out/${tag}_pmc_fetch $out/${tag}_pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/${tag}_prof.log 2>&1 || exit 2
find $out/${tag}_prof -name "*kernel_trace.csv" -delete
# the same command on one stream: per-kernel durations comparable with bench.py's roofline (measured on one stream)
rm -rf $out/${tag}_prof1
MSTG_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof1 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/${tag}_prof1.log 2>&1 || exit 2
find $out/${tag}_prof1 -name "*kernel_trace.csv" -delete
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $out/${tag}_pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $out/${tag}_pmc_write.log 2>&1 || exit 4
cd $GRAFT_REPO_ROOT
f=$(find $out/${tag}_pmc_fetch -name "*counter_collection.csv" | head -1)
w=$(find $out/${tag}_pmc_write -name "*counter_collection.csv" | head -1)
find $out/${tag}_pmc_fetch $out/${tag}_pmc_write -type f | head -20 > $out/${tag}_pmc_files.txt
python tools/pmc_traffic.py "$f" "$w" $out/${tag}_pmc_traffic.json; rc=$?
rm -rf $out/${tag}_pmc_fetch $out/${tag}_pmc_write
[ $rc -eq 0 ] || exit 5
echo refresh-done
