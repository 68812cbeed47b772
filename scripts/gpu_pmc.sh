# PMC passes (each its own run, kernel-trace only) for the two sweep kernels.  Outputs under gpurun_out/pmc/.
export TMPDIR=/tmp
REPO=$PWD
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
run() {  # name, counters...
  name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmc -o $name -- python $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline > $REPO/gpurun_out/pmc/$name.log 2>&1); echo "$name rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
run sq3 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_BUSY_CU_CYCLES
run spi SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_BAR_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN
run tcp TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run busy TA_BUSY_avr TCC_BUSY_avr GRBM_GUI_ACTIVE
python - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob('gpurun_out/pmc/*_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'rowshift' not in k and 'sweep' not in k: continue
        acc[k.split('(')[0][-36:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(os.path.basename(f)[:6], k[-28:], {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
