export TMPDIR=/tmp
REPO=$PWD
rm -rf gpurun_out/occ; mkdir -p gpurun_out/occ
for w in 4 5 8 10; do
  (cd /tmp && PD_ROW_WAVES=$w timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $REPO/gpurun_out/occ -o w$w -- python $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline > $REPO/gpurun_out/occ/w$w.log 2>&1); echo "w$w rc=$?"
done
python - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob('gpurun_out/occ/*_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'rowshift' not in k: continue
        acc[k.split('(')[0][-36:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        m = {c: sum(v)/len(v) for c, v in d.items()}
        dur = m['GRBM_GUI_ACTIVE'] / 8
        print(os.path.basename(f)[:4], k[-26:], "cycles/XCD %.0f  waves %d  resident waves/SIMD %.2f  VALU busy %.2f" % (dur, m['SQ_WAVES'], m['SQ_WAVE_CYCLES'] * 4 / (1024 * dur), m['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * dur)))
PY
