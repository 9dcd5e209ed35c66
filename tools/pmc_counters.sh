export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
run() { tag=$1; abl=$2; shift 2
  V1C_ABL=$abl rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc/$tag -o p -- python3 bench.py --no-cpu-baseline --traffic none --no-cold-extra --steps 3 --warmup 1 --no-condition --no-sustained --workload ${WL:-C2} > /dev/null 2>&1
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
abl=0
run a$abl $abl SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run b$abl $abl SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
run c$abl $abl SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64
run d$abl $abl TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run e$abl $abl GRBM_GUI_ACTIVE TA_TA_BUSY_sum SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
run f$abl $abl FETCH_SIZE
run g$abl $abl WRITE_SIZE
