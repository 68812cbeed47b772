export TMPDIR=/tmp
REPO=$PWD
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
run() { name=$1; shift
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmc -o $name -- python $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline > $REPO/gpurun_out/pmc/$name.log 2>&1); echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
run sq3 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE
python - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob('gpurun_out/pmc/*_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'rowshift' not in k: continue
        acc[k.split('(')[0][-36:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(os.path.basename(f)[:4], k[-28:], {c: round(sum(v)/len(v)/1e6,2) for c, v in d.items()})
PY
