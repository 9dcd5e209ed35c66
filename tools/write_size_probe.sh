# What WRITE_SIZE counts: a plain 100.66 MB device copy against the C2 launch (same number of destination bytes), and the EA write
# request mix of the C2 launch (32-byte against 64-byte requests).  bash tools/write_size_probe.sh   (GPU box)
export TMPDIR=/tmp
mkdir -p gpurun_out/wsp
cat > gpurun_out/wsp/copy.py <<'PY'
import torch
a = torch.randint(0, 255, (4096, 8192, 3), dtype=torch.uint8, device="cuda")
b = torch.empty_like(a)
for _ in range(4):
    b.copy_(a)
torch.cuda.synchronize()
PY
probe() { tag=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/wsp/$tag -o p -- python3 ${CMD} > /dev/null 2>&1
  f=$(find gpurun_out/wsp/$tag -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv,sys,collections
f,tag=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(list)
try:
    rows=list(csv.DictReader(open(f)))
except Exception as e:
    print(tag,'no data',e); sys.exit(0)
for r in rows:
    n=r['Kernel_Name']
    if 'k_ray' in n or 'copy' in n.lower():
        acc[(n[:48], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(tag, k, 'n=%d'%len(v), 'last', round(v[-1]))
PY
  rm -rf gpurun_out/wsp/$tag
}
CMD="gpurun_out/wsp/copy.py" probe copy_ws WRITE_SIZE
CMD="gpurun_out/wsp/copy.py" probe copy_req TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
CMD="bench.py --no-cpu-baseline --traffic none --no-cold-extra --steps 3 --warmup 1 --no-condition --workload C2" probe c2_ws WRITE_SIZE
CMD="bench.py --no-cpu-baseline --traffic none --no-cold-extra --steps 3 --warmup 1 --no-condition --workload C2" probe c2_req TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
CMD="bench.py --no-cpu-baseline --traffic none --no-cold-extra --steps 3 --warmup 1 --no-condition --workload C2A" probe c2a_ws WRITE_SIZE
