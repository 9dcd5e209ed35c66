#!/bin/bash
# SQ / LDS / TA / TCP / traffic counters of the dominant remap kernel for every workload (separate rocprofv3 --pmc
# passes, nothing else traced): bash tools/pmc_all.sh <outdir> [workloads...]   -> <outdir>/pmc_<W>.log
OUT=${1:-gpurun_out/pmc_all}; shift || true
WLS=${@:-C1 C2 C3 C4 C5}
mkdir -p $OUT
export TMPDIR=/tmp
for WL in $WLS; do
  echo "== $WL" | tee $OUT/pmc_$WL.log
  WL=$WL bash tools/pmc_counters.sh 2>&1 | tee -a $OUT/pmc_$WL.log
done
