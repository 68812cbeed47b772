# Round 6: the scheduler strategies on EVERY kernel of the library (variants allilp / allitilp: scripts/build_variants.sh with FILES = all sources):
# rocprofv3 kernel stats of three bench commands under each library, average microseconds per kernel side by side.
export TMPDIR=/tmp
REPO=$PWD
O=gpurun_out/r6/schedall; rm -rf $O; mkdir -p $O
run() { tag=$1; lib=$2; shift 2
  (cd /tmp && PD_LIB=${lib:+$REPO/planedepth_amd/lib/libpd_var_$lib.so} timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/$tag -o k -- \
     python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_ddp_step --launch eager "$@" > $REPO/$O/$tag.log 2>&1); echo "$tag rc=$?"
  cp $O/$tag/k_kernel_stats.csv $O/$tag.csv 2>/dev/null; rm -rf $O/$tag; }
for lib in "" allilp allitilp; do
  run default_${lib:-product} "$lib"
  run spec_${lib:-product} "$lib" --no_next_rows --warp_type homography_warp --mono_sides --automask --xz_levels 14
  run render_${lib:-product} "$lib" --no_next_rows --render_probability
  run colmap_${lib:-product} "$lib" --no_next_rows --warp_type homography_warp --colmap_pose
done
python - <<'PY'
import csv, glob, os, collections
O = 'gpurun_out/r6/schedall'
tab = collections.defaultdict(dict)
for f in sorted(glob.glob(O + '/*.csv')):
    cfg, lib = os.path.basename(f)[:-4].split('_', 1)
    for r in csv.DictReader(open(f)):
        n = r['Name']
        if n.startswith('void at::') or n.startswith('__amd') or 'at::native' in n: continue
        if int(r['Calls']) < 10: continue
        tab[(cfg, n.split('(')[0].replace('void pd::', '').replace('pd::', ''))][lib] = float(r['AverageNs']) / 1e3
with open(O + '/summary.txt', 'w') as out:
    for (cfg, k), d in sorted(tab.items(), key=lambda kv: (kv[0][0], -kv[1].get('product', 0))):
        p = d.get('product')
        if not p or p < 8: continue
        line = "%-8s %-70s product %8.1f us  allilp %8.1f (%+5.1f %%)  allitilp %8.1f (%+5.1f %%)" % (
            cfg, k[:70], p, d.get('allilp', 0), 100 * (d.get('allilp', p) / p - 1), d.get('allitilp', 0), 100 * (d.get('allitilp', p) / p - 1))
        print(line); out.write(line + "\n")
PY
