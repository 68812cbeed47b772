export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/r2
(cd /tmp && PD_UNI_CHUNK=8 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r2/prof_uni -o k -- python $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_next_rows --warp_type homography_warp --mono_pose > $REPO/gpurun_out/r2/prof_uni.log 2>&1); echo "rocprof rc=$?"
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r2/prof_uni/k_kernel_stats.csv")))
for r in rows[:5]: print(r["Name"][:50], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
