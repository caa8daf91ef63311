#!/bin/bash
# HERE (build container), after tools/gpu_refresh.sh ran on the GPU box: copy the summaries to be judged into profiles/.
#   usage: bash tools/collect_profiles.sh <tag>
tag=${1:-r02}
src=gpurun_out; dst=profiles
mkdir -p $dst
[ -f $src/${tag}_bench.log ] && grep "^\[kernels\]" $src/${tag}_bench.log | sed 's/^\[kernels\] //' > $dst/${tag}_bench_b32_256_kernel_table.txt && grep '^{' $src/${tag}_bench.log > $dst/${tag}_bench_b32_256.json
for v in prof prof1; do
  f=$(find $src/${tag}_$v -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] || continue
  name=$([ $v = prof ] && echo "" || echo "_one_stream")
  cp $f $dst/${tag}_bench_b32_256${name}_kernel_stats.csv
  python tools/profile_summary.py $f 8 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline  ($([ $v = prof ] && echo 'two streams (default)' || echo 'MSTG_STREAMS=0'); 2 warm-up + 5 timed + 1 instrumented step: per-step figures = totals / 8)" > $dst/${tag}_bench_b32_256${name}_summary.txt
done
[ -f $src/${tag}_pmc_traffic.json ] && cp $src/${tag}_pmc_traffic.json $dst/
[ -f $src/${tag}_c16_sq.txt ] && cp $src/${tag}_c16_sq.txt $dst/${tag}_sq_counters_c16.txt
[ -f $src/${tag}_c64_sq.txt ] && cp $src/${tag}_c64_sq.txt $dst/${tag}_sq_counters_c64_b8.txt
[ -f $src/${tag}_c64_bench.log ] && grep "^\[kernels\]" $src/${tag}_c64_bench.log | sed 's/^\[kernels\] //' > $dst/${tag}_bench_c64_b8_kernel_table.txt && grep '^{' $src/${tag}_c64_bench.log > $dst/${tag}_bench_c64_b8.json
[ -f $src/${tag}_cfg5_bench.log ] && grep "^\[kernels\]" $src/${tag}_cfg5_bench.log | sed 's/^\[kernels\] //' > $dst/${tag}_cfg5_fp16_1024_kernel_table.txt && grep '^{' $src/${tag}_cfg5_bench.log > $dst/${tag}_cfg5_fp16_1024.json
f=$(find $src/${tag}_cfg5_prof -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $dst/${tag}_cfg5_fp16_1024_kernel_stats.csv
[ -f $src/${tag}_cfg5_sq.txt ] && cp $src/${tag}_cfg5_sq.txt $dst/${tag}_sq_counters_cfg5.txt
ls -la $dst | grep ${tag}
