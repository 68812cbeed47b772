export TMPDIR=/tmp; REPO=$PWD; mkdir -p gpurun_out/r2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r2/n63 -o k -- python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --planes 49 --xz_levels 14 --automask > $REPO/gpurun_out/r2/n63.log 2>&1)
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/r2/n63/k_kernel_stats.csv')))[:12]:
    print(r['Name'][:90], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
