# Round 6 (as scripts/gpu_r5_configs.sh): every BASELINE configuration on the current build -> profiles/r06_configs.md, and BASELINE configs[3] AS SURVEY 8d (4)
# SPECIFIES IT (homography_warp, target views [r, -1, 1], automask, 49 xy + 14 xz planes with the decoder's normals /
# distances and the horizon mask of grid = linspace(-1, 1)): bench line, rocprofv3 kernel stats, FETCH / WRITE per kernel.
export TMPDIR=/tmp
REPO=$PWD
O=gpurun_out/r6/cfg; mkdir -p $O
SPEC="--warp_type homography_warp --mono_sides --automask --xz_levels 14"
rows=""
b() { name=$1; flags=$2; timeout 500 python bench.py --steps ${STEPS:-100} --warmup 20 --no_cpu_baseline --no_next_rows --no_ddp_step $flags > $O/c_$name.log 2>&1
      python - "$name" "$flags" $O/c_$name.log <<'PY'
import json, sys
name, flags, path = sys.argv[1:4]
line = [l for l in open(path) if l.startswith("{")]
if not line:
    print("| %s | `%s` | FAILED | | | | | |" % (name, flags)); sys.exit(0)
r = json.loads(line[-1]); k = r.get("kernels", {}); rp = r.get("roofline_path", {}); lp = r.get("launch_probe") or {}
print("| %s | `%s` | %.0f | %.4f | %s | %s | %s | %s / %s |" % (name, flags, r["value"], r["ms_per_step"], k.get("fwd_ms", ""), k.get("bwd_ms", ""),
      rp.get("frac", ""), lp.get("eager_images_per_sec", ""), lp.get("graph_images_per_sec", "")))
PY
}
{
echo "| configuration | bench.py flags | images/s | ms per step | forward ms (in step) | backward ms (in step) | path frac of 8 TB/s (algorithmic) | eager / graph images/s (probe) |"
echo "|---|---|---|---|---|---|---|---|"
b headline ""
b n63_xz "--xz_levels 14 --automask"
b batch12 "--batch 12"
b hr_384x1280 "--height 384 --width 1280 --batch 4"
b l1 "--no_mixture"
b homography_stereo "--warp_type homography_warp"
b homography_stereo_n63_xz_automask "--warp_type homography_warp --automask --xz_levels 14"
b homography_mono_pose "--warp_type homography_warp --mono_pose"
b homography_mono_sides_r4_workload "--warp_type homography_warp --mono_sides"
b configs3_as_specified "$SPEC"
b homography_colmap "--warp_type homography_warp --colmap_pose"
b render_probability "--render_probability"
} | tee $O/r06_configs.md
if [ -z "$NO_PROFILE" ]; then
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/stats -o spec -- python $REPO/bench.py --steps 50 --warmup 10 --no_cpu_baseline --no_next_rows --no_ddp_step $SPEC --launch eager > $REPO/$O/stats.log 2>&1); echo "stats rc=$?"
  cp $(ls $O/stats/*kernel_stats.csv | head -1) $O/r06_configs3_spec_kernel_stats.csv 2>/dev/null
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/$O/pmc -o $c -- python $REPO/bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_next_rows --no_ddp_step $SPEC --launch eager > $REPO/$O/pmc_$c.log 2>&1); echo "pmc $c rc=$?"
  done
  python - <<'PY' | tee gpurun_out/r6/cfg/r06_configs3_spec_traffic.txt
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r6/cfg/pmc/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'pd::' not in k: continue
        out[k.split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
tot = 0.0
for k, d in sorted(out.items()):
    m = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    n = len(next(iter(d.values())))
    mb = m.get('FETCH_SIZE', 0) * 2048 / 1e6 + m.get('WRITE_SIZE', 0) * 1024 / 1e6   # the guide's units: FETCH x 2 KiB (gfx950), WRITE x 1 KiB
    print('%-62s FETCH %8.0f WRITE %8.0f  ~%7.1f MB per launch  n=%d' % (k, m.get('FETCH_SIZE', 0), m.get('WRITE_SIZE', 0), mb, n))
PY
fi
