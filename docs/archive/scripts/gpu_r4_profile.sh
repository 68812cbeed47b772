# Round-4 profile set (everything that ends up under profiles/r04_*): kernel stats of the default bench and of the other
# configurations, FETCH/WRITE + SQ PMC passes of the headline kernels (each counter group its own run, kernel-trace only),
# the FETCH_SIZE calibration on the row-stream backward's access shape, the configuration table.
export TMPDIR=/tmp
REPO=$PWD
OUT=gpurun_out/r4/profile
if [ -z "$ONLY_CONFIGS" ]; then rm -rf $OUT; fi; mkdir -p $OUT/pmc
if [ -z "$ONLY_CONFIGS" ]; then
stats() {  # name flags...
  name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/$name -o k -- python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step "$@" > $REPO/$OUT/$name.log 2>&1); echo "stats $name rc=$?"
  cp $OUT/$name/k_kernel_stats.csv $OUT/r04_${name}_kernel_stats.csv
  rm -rf $OUT/$name
}
stats bench
stats n63_xz --xz_levels 14 --automask
stats hr --height 384 --width 1280 --batch 4
stats l1 --no_mixture
stats render --render_probability
stats homography_stereo --warp_type homography_warp
stats homography_mono_uniform --warp_type homography_warp --mono_pose
stats homography_mono_sides --warp_type homography_warp --mono_sides
stats homography_colmap --warp_type homography_warp --colmap_pose
pmc() {  # tag counters...
  tag=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/$OUT/pmc -o $tag -- python $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_next_rows --no_ddp_step > $REPO/$OUT/pmc_$tag.log 2>&1); echo "pmc $tag rc=$?"
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pmc sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH
# calibration: FETCH_SIZE / WRITE_SIZE of the probe kernels with KNOWN byte counts in the backward's access shape
cal() { tag=$1; shift; (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/$OUT/pmc -o cal_$tag -- $REPO/scripts/probes/stream_probe calib > $REPO/$OUT/cal_$tag.log 2>&1); echo "cal $tag rc=$?"; }
cal fetch FETCH_SIZE
cal write WRITE_SIZE
calf() { tag=$1; shift; (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/$OUT/pmc -o calf_$tag -- $REPO/scripts/probes/stream_probe Fwd > $REPO/$OUT/calf_$tag.log 2>&1); echo "calf $tag rc=$?"; }
calf fetch FETCH_SIZE
python - <<'PY'
import csv, glob, collections, json, os
OUT = 'gpurun_out/r4/profile'
def collect(pattern, keep):
    out = {}
    for f in sorted(glob.glob(OUT + '/pmc/' + pattern)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if not keep(k): continue
            acc[k.split('(')[0].replace('void pd::', '').replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in acc.items():
            out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in d.items()})
    return out
cal = collect('cal_*_counter_collection.csv', lambda k: 'stream<' in k)
B, N, H, W = 8, 49, 192, 640
taps, ctx = 2.0 * B * N * H * W * 4, 13.0 * B * H * W * 4
calib = {"known_bytes": {"tap_loads": taps, "context_staging": ctx, "stores": taps}, "kernels": {}}
for k, d in cal.items():
    stores = ', 3,' in k       # MODE = 3: loads + stores
    e = {"FETCH_SIZE": d.get("FETCH_SIZE"), "WRITE_SIZE": d.get("WRITE_SIZE")}
    if d.get("FETCH_SIZE"): e["FETCH_SIZE_bytes_per_KiB_counted"] = (taps + ctx) / (d["FETCH_SIZE"] * 1024)
    if d.get("WRITE_SIZE") and stores: e["WRITE_SIZE_bytes_per_KiB_counted"] = taps / (d["WRITE_SIZE"] * 1024)
    calib["kernels"][k] = e
calf = collect('calf_*_counter_collection.csv', lambda k: 'fwdrows<' in k)
for k, d in calf.items():   # the segment-stream forward's shape: 12-byte loads at 4-byte alignment
    known = taps + 6.0 * B * H * W * 4
    calib["kernels"][k] = {"FETCH_SIZE": d.get("FETCH_SIZE"), "known_read_bytes": known,
                           "FETCH_SIZE_bytes_per_KiB_counted": known / (d["FETCH_SIZE"] * 1024) if d.get("FETCH_SIZE") else None}
json.dump(calib, open(OUT + '/r04_fetch_calibration.json', 'w'), indent=1, sort_keys=True)
f12 = [e["FETCH_SIZE_bytes_per_KiB_counted"] for k, e in calib["kernels"].items() if e.get("FETCH_SIZE_bytes_per_KiB_counted") and 'stream<' in k]
f12 = sum(f12) / len(f12) if f12 else 2.0
ffs = [e["FETCH_SIZE_bytes_per_KiB_counted"] for k, e in calib["kernels"].items() if e.get("FETCH_SIZE_bytes_per_KiB_counted") and 'fwdrows<' in k]
ffs = sum(ffs) / len(ffs) if ffs else f12
f8 = json.load(open('profiles/r02_fetch_calibration.json'))['kernels']['P=1 U=2 mode=1 (L-)']['FETCH_SIZE_bytes_per_KiB_counted']
out = collect('[!c]*_counter_collection.csv', lambda k: 'row' in k or 'sweep' in k or 'fwdstream' in k)
for k, d in out.items():
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        f = f12 if 'rowstream' in k else (ffs if 'fwdstream' in k else f8)
        d['fetch_factor_used'] = f
        d['fetch_factor_source'] = ('profiles/r04_fetch_calibration.json (12-byte aligned loads, known byte counts)' if 'rowstream' in k
                                    else 'profiles/r04_fetch_calibration.json (12-byte loads at 4-byte alignment, one wave per segment: stream_probe Fwd)' if 'fwdstream' in k
                                    else 'profiles/r02_fetch_calibration.json (8-byte shifted loads, known byte counts)')
        d['hbm_read_bytes'] = d['FETCH_SIZE'] * 1024 * f
        d['hbm_write_bytes'] = d['WRITE_SIZE'] * 1024
        d['hbm_bytes'] = d['hbm_read_bytes'] + d['hbm_write_bytes']
json.dump(out, open(OUT + '/r04_pmc_summary.json', 'w'), indent=1, sort_keys=True)
for k, d in out.items():
    print(k, {c: round(v) for c, v in d.items() if c in ('hbm_read_bytes', 'hbm_write_bytes', 'hbm_bytes', 'SQ_INSTS_VALU', 'SQ_WAVES', 'FETCH_SIZE', 'WRITE_SIZE')})
print("calibration", json.dumps(calib["kernels"], indent=1))
PY
rm -rf $OUT/pmc
fi   # ONLY_CONFIGS
# configuration table (100 timed steps after 20: a fresh process runs its first ~50 steps slower)
b() { name=$1; shift; timeout 400 python bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_next_rows --no_ddp_step "$@" > $OUT/cfg_$name.log 2>&1; echo "| $name | \`$*\` | $(python - $OUT/cfg_$name.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
k = d.get('kernels', {})
r = d.get('roofline') or d.get('roofline_general') or {}
print("%.0f | %.4f | %s | %s | %s" % (d['value'], d['ms_per_step'], k.get('fwd_ms'), k.get('bwd_ms'), r.get('frac')))
PY
) |"; }
{
echo "| configuration | bench.py flags | images/s | ms per step | forward ms (in step) | backward ms (in step) | roofline.frac (dominant kernel) |"
echo "|---|---|---|---|---|---|---|"
b headline
PD_SWEEP_IMPL=4 b headline_rowshift_kernels_PD_SWEEP_IMPL_4
b n63_xz --xz_levels 14 --automask
b batch12 --batch 12
b hr_384x1280 --height 384 --width 1280 --batch 4
b l1 --no_mixture
b render_probability --render_probability
b homography_stereo --warp_type homography_warp
b homography_mono_pose --warp_type homography_warp --mono_pose
b homography_mono_sides --warp_type homography_warp --mono_sides
b homography_colmap --warp_type homography_warp --colmap_pose
} > $OUT/r04_configs.md
cat $OUT/r04_configs.md
find $OUT -name "*.log" -size +50k -delete
