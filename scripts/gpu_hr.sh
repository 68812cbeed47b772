# HR configuration (384x1280, batch 4) under different waves-per-row-workgroup settings
mkdir -p gpurun_out
for w in 4 6 8 5; do
  PD_ROW_WAVES=$w timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows --batch 4 --height 384 --width 1280 > gpurun_out/hr_$w.log 2>&1
  echo "waves=$w $(grep -oE '"value": [0-9.]*' gpurun_out/hr_$w.log | head -1) $(grep -oE '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/hr_$w.log | head -1)"
done
