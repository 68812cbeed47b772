# FETCH_SIZE / WRITE_SIZE of the plane-uniform kernels inside the --mono_sides step (the scratch's share of the HBM traffic).
export TMPDIR=/tmp
REPO=$PWD
OUT=gpurun_out/r4/sidespmc; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/$OUT -o $c -- python $REPO/bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_next_rows --no_ddp_step --warp_type homography_warp --mono_sides --launch eager > $REPO/$OUT/$c.log 2>&1); echo "pmc $c rc=$?"
done
python - <<'PY'
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r4/sidespmc/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'pd::' not in k: continue
        out[k.split('(')[0][-48:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(out.items()):
    print('%-50s' % k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, 'n=%d' % len(next(iter(d.values()))))
PY
