#!/bin/bash
# PMC counters of the hot kernel for one engine build: bash tools/pmc_one.sh <workload> <lib.so> COUNTER...
export TMPDIR=/tmp
WL=$1; L=$2; shift 2
D=$(mktemp -d /tmp/pmc.XXXX)
V1C_LIB=$L rocprofv3 --pmc "$@" --output-format csv -d $D -o p -- python3 bench.py --no-cpu-baseline --traffic none --no-cold-extra --steps 3 --warmup 1 --no-condition --workload $WL > /dev/null 2>&1
f=$(find $D -name '*counter_collection.csv' | head -1)
python3 - "$f" "$L" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_ray' in r['Kernel_Name'] or 'k_remap' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2].split('/')[-1], {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
rm -rf $D
