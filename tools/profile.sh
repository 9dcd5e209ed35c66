#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the bench command, then separate PMC
# passes for FETCH_SIZE and WRITE_SIZE (rocprofv3 cannot hold both in one pass on gfx950).
# Usage: bash tools/profile.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py "$@" > $OUT/bench_line.json 2> $OUT/bench_stderr.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --no-cpu-baseline --traffic none --no-cold-extra "$@" > $OUT/bench_traced.log 2>&1
find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/trace -name '*kernel_trace.csv' | head -1 | xargs -I{} sh -c 'head -1 {} > '$OUT'/kernel_trace_head.csv; grep -E "k_ray|k_remap|k_tile" {} | head -100 >> '$OUT'/kernel_trace_head.csv'
for k in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $k --output-format csv -d $OUT/pmc_$k -o pmc -- python3 bench.py --no-cpu-baseline --traffic none --no-cold-extra --no-condition --steps 5 --warmup 2 "$@" > $OUT/bench_pmc_$k.log 2>&1
  f=$(find $OUT/pmc_$k -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { head -1 "$f" > $OUT/pmc_$k.csv; grep -E "k_ray|k_remap" "$f" | head -40 >> $OUT/pmc_$k.csv; }
  rm -rf $OUT/pmc_$k
done
rm -rf $OUT/trace
python3 tools/traffic.py $OUT "$@" > $OUT/traffic.json
cat $OUT/traffic.json
