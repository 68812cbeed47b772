for w in 512 640 768 1024; do
  python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows --width $w > gpurun_out/w_$w.log 2>&1
  echo "W=$w $(grep -oE '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/w_$w.log | head -1)"
done
