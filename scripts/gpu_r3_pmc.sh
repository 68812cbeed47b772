# Round 3: SQ + HBM counters of the sweep kernels inside the bench step (separate --pmc passes, kernel trace only).
# Outputs gpurun_out/r3/pmc/*, summary gpurun_out/r3/pmc_summary.json
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/r3/pmc
run() {  # tag counters...
  tag=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/gpurun_out/r3/pmc -o $tag -- python $REPO/bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_next_rows $CFG > $REPO/gpurun_out/r3/pmc/$tag.log 2>&1); echo "$tag rc=$?"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH
python - <<'PY'
import csv, glob, collections, os, json
out = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/r3/pmc/*_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'row' not in k and 'sweep' not in k: continue
        acc[k.split('(')[0].replace('void pd::', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        for c, v in d.items():
            out[k][c] = sum(v) / len(v)
json.dump(out, open('gpurun_out/r3/pmc_summary.json', 'w'), indent=1, sort_keys=True)
for k, d in out.items():
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
find gpurun_out/r3/pmc -name "*.csv" -size +200k -delete
