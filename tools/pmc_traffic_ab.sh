#!/bin/bash
# HBM traffic of one workload under two builds: FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, KiB; FETCH_SIZE x 2 on gfx950:
# /opt/skills/guides/MI355X_MICROARCH.md) summed over the remap kernel's dispatches of a short bench run, per step.
#   bash tools/pmc_traffic_ab.sh <workload> default path/to/variant.so
WL=$1; shift
export TMPDIR=/tmp
for L in "$@"; do
  [ "$L" = default ] && unset V1C_LIB || export V1C_LIB=$L
  for C in FETCH_SIZE WRITE_SIZE; do
    D=$(mktemp -d)
    rocprofv3 --pmc $C --output-format csv -d $D -o pmc -- python3 bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --traffic none --no-cold-extra --no-condition --no-sustained > /dev/null 2>&1
    python3 - $D $C "$L" $WL <<'PY'
import csv, sys, pathlib
d, c, lib, wl = sys.argv[1:5]
tot = {}
for f in pathlib.Path(d).rglob("*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and ("k_ray" in r["Kernel_Name"] or "k_remap" in r["Kernel_Name"]):
            k = r["Kernel_Name"].split("(")[0]
            tot.setdefault(k, [0.0, 0])
            tot[k][0] += float(r["Counter_Value"]); tot[k][1] += 1
for k, (v, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{wl} {lib} {c} {k}: {v / n * 1024 * (2 if c == 'FETCH_SIZE' else 1) / 1e6:.1f} MB per dispatch ({n} dispatches)")
PY
    rm -rf $D
  done
done
