# per-kernel averages of the smaller operators (scripts/op_times.py)
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/prof_ops
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_ops -o k -- python $REPO/scripts/op_times.py > $REPO/gpurun_out/prof_ops/run.log 2>&1); echo "rc=$?"
tail -3 gpurun_out/prof_ops/run.log | cut -c1-200
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_ops/k_kernel_stats.csv")))
for r in rows:
    if "pd::" in r["Name"]: print(r["Name"][:84].ljust(84), r["Calls"], round(float(r["AverageNs"])/1000,1))
PY
