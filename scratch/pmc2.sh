export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
run() { tag=$1; abl=$2; shift 2
  V1C_ABL=$abl rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc/$tag -o p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  f=$(find gpurun_out/pmc/$tag -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv,sys,collections
f,tag=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_ray' in r['Kernel_Name'] or 'k_remap' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(tag, {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
  rm -rf gpurun_out/pmc/$tag
}
for abl in 0 256 512; do
  V1C_ABL=$abl python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('abl=$abl', 'kernel_ms', d['roofline']['kernel_ms'], 'Gpx/s', d['value']/1e3)"
  run c$abl $abl SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
done
