#!/bin/bash
# Everything a round's profiles/ directory holds, for every workload (run on the GPU box through gpurun):
#   <W>_bench_line.json    un-profiled bench line (L3-cold rotation; roofline.traffic measured live by two --pmc child passes)
#   <W>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command (the dominant kernel's average must agree
#                          with roofline.kernel_ms)
#   pmc_<W>.log            SQ / LDS / TA / TCP counter passes (tools/pmc_counters.sh)
#   traffic.json           HBM bytes per step of all workloads (what bench.py falls back to when rocprofv3 is unavailable)
# bash tools/profile_round.sh <outdir> [workloads...]
set -u
OUT=${1:-gpurun_out/prof_round}; shift || true
WLS=${@:-C1 C2 C3 C4 C5}
mkdir -p $OUT
export TMPDIR=/tmp
for W in $WLS; do
  EXTRA="--no-cpu-baseline --no-cold-extra"; [ "$W" = C2 ] && EXTRA=""
  python3 bench.py --workload $W $EXTRA > $OUT/${W}_bench_line.json 2> /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -o trace -- python3 bench.py --no-cpu-baseline --traffic none --no-cold-extra --workload $W --steps 200 --warmup 20 --no-sustained > /dev/null 2>&1
  find $OUT/trace_$W -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${W}_kernel_stats.csv
  rm -rf $OUT/trace_$W
  echo "== $W" > $OUT/pmc_$W.log
  WL=$W bash tools/pmc_counters.sh >> $OUT/pmc_$W.log 2>&1
  echo "$W done"
done
python3 - $OUT <<'PY'
import json, sys, pathlib
out = pathlib.Path(sys.argv[1]); t = {}
for f in sorted(out.glob('*_bench_line.json')):
    try:
        d = json.loads(f.read_text().strip().splitlines()[-1])
    except Exception:
        continue
    r = d['roofline']
    if r.get('traffic'):
        t[f.name.split('_')[0]] = {'hbm_bytes_per_step': r['traffic'], 'tag': out.name, **{k: v for k, v in r.get('traffic_detail', {}).items() if k != 'hbm_bytes_per_step'}}
    print(f.name.split('_')[0], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'traffic', r.get('traffic'), 'alg', r['algorithmic_bytes_per_launch'])
(out / 'traffic.json').write_text(json.dumps(t, indent=1))
PY
head -4 $OUT/*_kernel_stats.csv | cut -c1-200
