# Round-6 profile set (as scripts/gpu_r5_profile.sh) of the HEADLINE workload (profiles/r06_*): rocprofv3 kernel stats of the default bench command, FETCH / WRITE /
# SQ PMC passes of its kernels (each counter group its own run, kernel-trace only).  The FETCH_SIZE factors are round 4's
# calibration on the same two access shapes (profiles/r04_fetch_calibration.json: the load shapes did not change).
export TMPDIR=/tmp
REPO=$PWD
OUT=gpurun_out/r6/profile; rm -rf $OUT; mkdir -p $OUT/pmc
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/bench -o k -- python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --launch eager > $REPO/$OUT/bench.log 2>&1); echo "stats rc=$?"
cp $OUT/bench/k_kernel_stats.csv $OUT/r06_bench_kernel_stats.csv; rm -rf $OUT/bench
pmc() { tag=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/$OUT/pmc -o $tag -- python $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_next_rows --no_ddp_step --launch eager > $REPO/$OUT/pmc_$tag.log 2>&1); echo "pmc $tag rc=$?"; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pmc sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
python - <<'PY'
import csv, glob, collections, json
OUT = 'gpurun_out/r6/profile'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(OUT + '/pmc/*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not ('rowstream' in k or 'fwdstream' in k): continue
        acc[k.split('(')[0].replace('void pd::', '').replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
cal = json.load(open('profiles/r04_fetch_calibration.json'))['kernels']
f12 = [e["FETCH_SIZE_bytes_per_KiB_counted"] for k, e in cal.items() if e.get("FETCH_SIZE_bytes_per_KiB_counted") and 'stream<' in k]
ffs = [e["FETCH_SIZE_bytes_per_KiB_counted"] for k, e in cal.items() if e.get("FETCH_SIZE_bytes_per_KiB_counted") and 'fwdrows<' in k]
f12, ffs = sum(f12) / len(f12), sum(ffs) / len(ffs)
out = {}
for k, d in acc.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
        f = f12 if 'rowstream' in k else ffs
        e.update(fetch_factor_used=f, fetch_factor_source='profiles/r04_fetch_calibration.json (same access shape)',
                 hbm_read_bytes=e['FETCH_SIZE'] * 1024 * f, hbm_write_bytes=e['WRITE_SIZE'] * 1024)
        e['hbm_bytes'] = e['hbm_read_bytes'] + e['hbm_write_bytes']
    out[k] = e
json.dump(out, open(OUT + '/r06_pmc_summary.json', 'w'), indent=1, sort_keys=True)
for k, e in out.items():
    print(k, {c: round(v) for c, v in e.items() if c in ('hbm_read_bytes', 'hbm_write_bytes', 'hbm_bytes', 'SQ_INSTS_VALU', 'SQ_WAVES', 'SQ_ACTIVE_INST_VALU', 'SQ_BUSY_CYCLES')})
PY
rm -rf $OUT/pmc; find $OUT -name "*.log" -size +50k -delete
