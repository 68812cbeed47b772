# Issue/stall breakdown of the sweep kernels (SQ/TCC/TCP counters, separate passes) -> gpurun_out/pmc2/summary.json
# CFG="--warp_type homography_warp" selects another bench configuration
export TMPDIR=/tmp
REPO=$PWD
rm -rf gpurun_out/pmc2; mkdir -p gpurun_out/pmc2
pass() { name=$1; shift; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmc2 -o $name -- python $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_next_rows $CFG > $REPO/gpurun_out/pmc2/$name.log 2>&1); echo "$name rc=$?"; }
pass a SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
pass b SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass d FETCH_SIZE
pass e WRITE_SIZE
pass f TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_REQ_sum TCC_EA0_ATOMIC_sum TCC_TAG_STALL_sum
pass g TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
pass c SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob('gpurun_out/pmc2/*_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'rowshift' not in k and 'sweep_' not in k: continue
        acc[k.split('(')[0].replace('void pd::', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in d.items()})
json.dump(out, open('gpurun_out/pmc2/summary.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
