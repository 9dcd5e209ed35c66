#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py for every workload: bash tools/profile_all.sh <tag>
# -> gpurun_out/prof_<tag>/<workload>_{kernel_stats.csv,bench_line.json}
set -u
TAG=${1:-all}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for W in C1 C2 C3 C4 C5; do
  python3 bench.py --workload $W --no-cpu-baseline --no-cold-extra > $OUT/${W}_bench_line.json 2> /dev/null   # roofline.traffic measured live (two --pmc child passes)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -o trace -- python3 bench.py --no-cpu-baseline --traffic none --no-cold-extra --workload $W --steps 40 --warmup 5 > /dev/null 2>&1
  find $OUT/trace_$W -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${W}_kernel_stats.csv
  rm -rf $OUT/trace_$W
done
head -3 $OUT/*_kernel_stats.csv
