#!/bin/bash
# bench lines (no CPU baseline) of the workloads given as arguments: bash tools/quick_bench.sh C2 C3 C5
for WL in "$@"; do
  python3 bench.py --no-cpu-baseline --traffic none --no-cold-extra --workload $WL 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$WL', 'ms/step', d['ms_per_step'], 'Gpx/s', round(d['value']/1e3,1), 'frac', d['roofline']['frac'], 'parity', d.get('parity_vs_oracle'))
"
done
