#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): everything the committed profiles/ of a round are made from.
#   usage: bash tools/gpu_refresh.sh <tag> [part ...]      parts: bench prof prof1 pmc sq c64 cfg5   (default: all)
# Outputs under gpurun_out/<tag>_*; copy what is to be judged into profiles/ afterwards (tools/collect_profiles.sh).
set -o pipefail
tag=${1:-r02}; shift
parts=${@:-bench prof prof1 pmc sq c64 cfg5}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
has() { [[ " $parts " == *" $1 "* ]]; }
cd $GRAFT_REPO_ROOT
if has bench; then
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --kernel-table > $out/${tag}_bench.log 2>&1 || exit 1
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then   # the default two-stream schedule
  rm -rf $out/${tag}_prof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/${tag}_prof.log 2>&1 || exit 2
  find $out/${tag}_prof -name "*kernel_trace.csv" -delete
fi
if has prof1; then  # the same command on one stream: per-kernel durations comparable with bench.py's roofline (measured on one stream)
  rm -rf $out/${tag}_prof1
  MSTG_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof1 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/${tag}_prof1.log 2>&1 || exit 2
  find $out/${tag}_prof1 -name "*kernel_trace.csv" -delete
fi
if has pmc; then    # HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X guide)
  rm -rf $out/${tag}_pmc_fetch $out/${tag}_pmc_write
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $out/${tag}_pmc_fetch.log 2>&1 || exit 3
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $out/${tag}_pmc_write.log 2>&1 || exit 4
  f=$(find $out/${tag}_pmc_fetch -name "*counter_collection.csv" | head -1)
  w=$(find $out/${tag}_pmc_write -name "*counter_collection.csv" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_traffic.py "$f" "$w" $out/${tag}_pmc_traffic.json || exit 5
  rm -rf $out/${tag}_pmc_fetch $out/${tag}_pmc_write
fi
cd $GRAFT_REPO_ROOT
if has sq; then     # MFMA / LDS / wait counters of every kernel of the headline step (channels=16)
  bash tools/gpu_pmc_bench.sh ${tag}_c16 || exit 6
fi
if has c64; then    # the class-default width: where the "40 % MFMA on 3x3 conv" target is to be read
  timeout -k 10 300 python bench.py --channels 64 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --kernel-table > $out/${tag}_c64_bench.log 2>&1 || exit 7
  bash tools/gpu_pmc_bench.sh ${tag}_c64 --channels 64 --batch 8 || exit 8
fi
if has cfg5; then   # fp16 inference, 1024x1024 batch 64
  timeout -k 10 300 python bench.py --config 5 --steps 10 --warmup 3 --kernel-table > $out/${tag}_cfg5_bench.log 2>&1 || exit 9
  cd /tmp
  rm -rf $out/${tag}_cfg5_prof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_cfg5_prof -o run -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline > $out/${tag}_cfg5_prof.log 2>&1 || exit 10
  find $out/${tag}_cfg5_prof -name "*kernel_trace.csv" -delete
  cd $GRAFT_REPO_ROOT
  bash tools/gpu_pmc_bench.sh ${tag}_cfg5 --config 5 || exit 11
fi
echo refresh-done
