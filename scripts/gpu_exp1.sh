# Experiment session: parity tests on the new build, soffset probe, width sweep (segment balance), headline bench.
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
./scripts/probes/soff_probe
for w in 512 640 768 1024 1280; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --width $w > gpurun_out/w_$w.log 2>&1
  echo "W=$w $(grep -o '"value": [0-9.]*' gpurun_out/w_$w.log | head -1) $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/w_$w.log)"
done
timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --automask > gpurun_out/b_auto.log 2>&1; grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/b_auto.log
timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_plane_grad > gpurun_out/b_nopg.log 2>&1; grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/b_nopg.log
