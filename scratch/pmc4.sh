export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
run() { tag=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc/$tag -o p -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload C4 > /dev/null 2>&1
  f=$(find gpurun_out/pmc/$tag -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv,sys,collections
f,tag=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_ray_lin3_tile' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(tag, {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
  rm -rf gpurun_out/pmc/$tag
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
run c SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
run d TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run e GRBM_GUI_ACTIVE TA_TA_BUSY_sum
