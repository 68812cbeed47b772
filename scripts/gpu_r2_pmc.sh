# SQ counters of the sweep kernels: product (row-quad) vs PD_SWEEP_IMPL=4 (one pixel per lane).  Outputs gpurun_out/r2/pmc_*
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/r2/pmc
run() {  # tag impl counters...
  tag=$1; impl=$2; shift 2
  (cd /tmp && PD_SWEEP_IMPL=$impl timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/gpurun_out/r2/pmc -o $tag -- python $REPO/bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_next_rows > $REPO/gpurun_out/r2/pmc/$tag.log 2>&1); echo "$tag rc=$?"
}
for impl in 0 4; do
  run sq1_$impl $impl SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
  run sq2_$impl $impl SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
  run sq3_$impl $impl SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH
done
python - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob('gpurun_out/r2/pmc/*_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'row' not in k: continue
        acc[k.split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(os.path.basename(f)[:6], k[-34:], {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
