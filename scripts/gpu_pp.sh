# post-process kernels: parity tests, then per-kernel times of the row kernels (default) and the general gathers (PD_PP_ROWS=0)
timeout 600 python -m pytest tests -m gpu -q -k "post_process" 2>&1 | tail -2
PD_PP_ROWS=0 timeout 600 python -m pytest tests -m gpu -q -k "post_process" 2>&1 | tail -1
for l in 1 0; do
  PD_PP_ROWS=$l bash scripts/gpu_prof_next.sh > /dev/null
  echo "PD_PP_ROWS=$l"
  python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_next/k_kernel_stats.csv")))
for r in rows:
    if "warp_" in r["Name"]: print("  ", r["Name"][:62], r["Calls"], round(float(r["AverageNs"])/1000,1))
PY
done
