# per-kernel times of the --mono_sides step: rocprofv3 kernel trace, summary to gpurun_out/sides_stats$TAG.txt
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out; rm -rf gpurun_out/sprof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/sprof -o k -- python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --warp_type homography_warp --mono_sides $EXTRA > $REPO/gpurun_out/sprof.log 2>&1); echo "rc=$?"
python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/sprof/k_kernel_stats.csv')))
with open('gpurun_out/sides_stats$TAG.txt', 'w') as f:
    for r in rows[:9]:
        line = "%-90s calls %5s avg_us %9.1f pct %5s" % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage'])
        print(line); f.write(line + "\n")
PY
rm -rf gpurun_out/sprof
