#!/bin/bash
# What does dealing the steps to two HIP streams cost, and where?  GPU-side kernel durations (rocprofv3 --kernel-trace) next to the
# bench's per-step times for --streams 1 and 2: if the kernels take as long as with one stream but the step time doubles, the loss is
# the host's (stream switching per step), not the GPU's.   bash tools/streams_probe.sh [workload]
export TMPDIR=/tmp
WL=${1:-C2}
for S in 1 2; do
  D=$(mktemp -d /tmp/streams.XXXX)
  rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 bench.py --no-cpu-baseline --traffic none --no-cold-extra --no-sustained --workload $WL --steps 400 --warmup 20 --streams $S > $D/line.json 2>/dev/null
  f=$(find $D -name '*kernel_trace.csv' | head -1)
  python3 - "$f" "$D/line.json" "$S" <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_ray' in r['Kernel_Name']]
rows = rows[-400:]
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
st = sorted(int(r['Start_Timestamp']) for r in rows)
span = (max(int(r['End_Timestamp']) for r in rows) - st[0]) / 1e3 / len(rows)
line = json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
print(f"streams {sys.argv[3]}: kernel avg {sum(dur)/len(dur):.1f} us (min {min(dur):.1f}), launch-to-launch {span:.1f} us, bench ms_per_step {line['ms_per_step']} kernel_ms {line['roofline']['kernel_ms']}")
PY
  rm -rf $D
done
