#!/bin/bash
# A/B of engine builds under the bench's L3-cold protocol, REPS interleaved repetitions (process-to-process
# noise is several per cent: buffer placement, clocks), min and median printed:
#   bash tools/ab.sh "C2 C3" default path/to/lib.so "default:V1C_XCD_STRIPS=4" ...
# an arm is <lib>[:ENV=val[,ENV=val...]]; "default" = the in-tree build (vr180_convert_amd/csrc)
WLS=$1; shift
REPS=${REPS:-3}
TMP=$(mktemp)
for r in $(seq $REPS); do
  for ARM in "$@"; do
    L=${ARM%%:*}; E=""; [ "$ARM" != "$L" ] && E=$(echo "${ARM#*:}" | tr ',' ' ')
    for WL in $WLS; do
      # (environment switches exist only in the -DV1C_TUNING build: "default" with switches -> the tuning twin)
      if [ "$L" = default ] && [ -z "$E" ]; then LIBENV=""; elif [ "$L" = default ] || [ "$L" = tuning ]; then LIBENV="V1C_LIB=vr180_convert_amd/csrc/libvr180remap_tuning.so"; else LIBENV="V1C_LIB=$L"; fi
      env $LIBENV $E python3 bench.py --no-cpu-baseline --traffic none --no-cold-extra --workload $WL ${ARGS:-} 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$ARM', '$WL', d['roofline']['kernel_ms'])
" >> $TMP
    done
  done
done
python3 - $TMP <<'PY'
import sys, collections, statistics
acc = collections.defaultdict(list)
for l in open(sys.argv[1]):
    arm, wl, ms = l.split()
    acc[(wl, arm)].append(float(ms))
for (wl, arm), v in sorted(acc.items()):
    print('%-4s %-60s min %.4f  med %.4f  (n=%d: %s)' % (wl, arm, min(v), statistics.median(v), len(v), ' '.join('%.4f' % x for x in v)))
PY
rm -f $TMP
