#!/bin/bash
# GPU-side duration of the hot kernel for one or more engine builds (A/B tests, ablations):
# rocprofv3 --kernel-trace of a short bench run, average over the timed launches.
# bench.py's own per-step events include host launch latency (~45 us/step) and cannot resolve
# kernels shorter than that.
# Usage: bash tools/ktime.sh <workload> <lib.so> [<lib.so> ...]
export TMPDIR=/tmp
WL=$1; shift
for L in "$@"; do
  D=$(mktemp -d /tmp/ktime.XXXX)
  V1C_LIB=$L rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 bench.py --no-cpu-baseline --traffic none --no-cold-extra --workload $WL --steps 40 --warmup 5 --no-sustained > /dev/null 2>&1
  f=$(find $D -name '*kernel_trace.csv' | head -1)
  python3 - "$f" "$L" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'k_ray' in n or 'k_remap' in n:
        acc[n.split('(')[0][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in acc.items():
    v = v[5:] if len(v) > 10 else v
    print(f"{sys.argv[2].split('/')[-1]:40s} {k:60s} n={len(v):3d} avg {sum(v)/len(v):8.2f} us  min {min(v):8.2f} us")
PY
  rm -rf $D
done
