#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the bench command, then PMC passes.
# Usage: bash tools_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --no-cpu-baseline "$@" > $OUT/bench_traced.log 2>&1
find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/trace -name '*kernel_trace.csv' | head -1 | xargs -I{} sh -c 'head -1 {} > '$OUT'/kernel_trace_head.csv; grep k_remap {} | head -200 >> '$OUT'/kernel_trace_head.csv'
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $OUT/bench_pmc_write.log 2>&1
for k in fetch write; do
  f=$(find $OUT/pmc_$k -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { head -1 "$f" > $OUT/pmc_$k.csv; grep k_remap "$f" | head -40 >> $OUT/pmc_$k.csv; }
done
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
