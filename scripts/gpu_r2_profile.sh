# Round-2 profile set (everything that ends up under profiles/r02_*):
#   kernel stats of the default bench and of the homography configurations, FETCH/WRITE + SQ PMC passes of the headline
#   kernels (each counter group its own run, kernel-trace only), the secondary-configuration table.
export TMPDIR=/tmp
REPO=$PWD
OUT=gpurun_out/r2/profile
rm -rf $OUT; mkdir -p $OUT
stats() {  # name flags...
  name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/$name -o k -- python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step "$@" > $REPO/$OUT/$name.log 2>&1); echo "stats $name rc=$?"
  cp $OUT/$name/k_kernel_stats.csv $OUT/r02_${name}_kernel_stats.csv
}
stats bench
stats homography_stereo --warp_type homography_warp
stats homography_mono_uniform --warp_type homography_warp --mono_pose
stats homography_colmap --warp_type homography_warp --colmap_pose
stats homography_stereo_general --warp_type homography_warp --general_stereo
stats homography_mono_sides --warp_type homography_warp --mono_sides
BENCH_FLAGS=""
pmc() {  # tag counters...
  tag=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/$OUT/pmc -o $tag -- python $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_next_rows --no_ddp_step $BENCH_FLAGS > $REPO/$OUT/pmc_$tag.log 2>&1); echo "pmc $tag rc=$?"
}
BENCH_FLAGS="--warp_type homography_warp --mono_pose"
pmc uni_fetch FETCH_SIZE
pmc uni_write WRITE_SIZE
BENCH_FLAGS="--warp_type homography_warp --colmap_pose"
pmc gen_fetch FETCH_SIZE
pmc gen_write WRITE_SIZE
BENCH_FLAGS=""
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pmc sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
python - <<'PY'
import csv, glob, collections, json, os
out = {}
for f in sorted(glob.glob('gpurun_out/r2/profile/pmc/*_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'row' not in k and 'sweep' not in k and 'uniform' not in k: continue
        acc[k.split('(')[0].replace('void pd::', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in d.items()})
cal = json.load(open('profiles/r02_fetch_calibration.json'))['kernels']
f8 = cal['P=1 U=2 mode=1 (L-)']['FETCH_SIZE_bytes_per_KiB_counted']      # 8-byte shifted loads (row-shift kernels)
f12 = cal['P=2 U=2 mode=1 (L-)']['FETCH_SIZE_bytes_per_KiB_counted']     # 12-byte loads (row-quad forward, Q = 2)
for k, d in out.items():
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        f = f12 if 'rowquad' in k else (f8 if 'rowshift' in k else 2.0)
        d['fetch_factor_used'] = f
        d['fetch_factor_source'] = ('profiles/r02_fetch_calibration.json (known byte counts, this access width)'
                                    if 'row' in k else 'MI355X_MICROARCH.md gfx950 correction (x2); 4-byte gathers are not calibrated')
        d['hbm_read_bytes'] = d['FETCH_SIZE'] * 1024 * f     # calibrated on known byte counts in this access shape
        d['hbm_write_bytes'] = d['WRITE_SIZE'] * 1024        # WRITE_SIZE counts 2-4 % above the algorithmic bytes: partial lines, real traffic
        d['hbm_bytes'] = d['hbm_read_bytes'] + d['hbm_write_bytes']
json.dump(out, open('gpurun_out/r2/profile/r02_pmc_summary.json', 'w'), indent=1, sort_keys=True)
for k, d in out.items():
    print(k, {c: round(v) for c, v in d.items() if c in ('hbm_read_bytes', 'hbm_write_bytes', 'hbm_bytes', 'SQ_INSTS_VALU', 'SQ_WAVES')})
PY
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step "$@" > $OUT/t_$name.log 2>&1; echo "| $name | \`$*\` | $(grep -o '"value": [0-9.]*' $OUT/t_$name.log | head -1 | cut -d' ' -f2) | $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' $OUT/t_$name.log | head -1) |"; }
{
echo "| config | bench.py flags | images/s | sweep kernels inside the step |"
echo "|---|---|---|---|"
b headline
b n63_xz_automask --planes 49 --xz_levels 14 --automask
b batch12 --batch 12
b hr_384x1280 --batch 4 --height 384 --width 1280
b l1_no_mixture --no_mixture
b render_probability --render_probability
PD_SWEEP_IMPL=1 b render_probability_general_kernels --render_probability
b render_probability_homography_mono_sides --render_probability --warp_type homography_warp --mono_sides
b homography_stereo_49 --warp_type homography_warp
b homography_mono_f8_49 --warp_type homography_warp --mono_pose
b homography_mono_f8_49_automask --warp_type homography_warp --mono_pose --automask
b homography_colmap_49 --warp_type homography_warp --colmap_pose
b homography_stereo_49_general_kernels --warp_type homography_warp --general_stereo
b homography_mono_sides_r_m1_p1_49 --warp_type homography_warp --mono_sides
b homography_mono_sides_63_automask --warp_type homography_warp --mono_sides --xz_levels 14 --automask
PD_TORCH_HOMOGRAPHY=1 b homography_mono_f8_49_torch_algebra --warp_type homography_warp --mono_pose
PD_UNI_FUSED=1 b homography_mono_f8_49_fused_bwd_optin --warp_type homography_warp --mono_pose
PD_SWEEP_IMPL=4 b rows1_headline
PD_SWEEP_IMPL=2 b fast_rows_optin
PD_SWEEP_IMPL=3 b tile_backward_homography_stereo --warp_type homography_warp --general_stereo
b homography_mono_sides_one_node_per_view --warp_type homography_warp --mono_sides --per_view_nodes
} | tee $OUT/r02_configs_table.md
for n in bench homography_stereo homography_mono_uniform homography_colmap homography_stereo_general homography_mono_sides; do echo "== $n"; head -5 $OUT/r02_${n}_kernel_stats.csv | cut -d, -f1-5 | cut -c1-140; done
