# stereo side of homography_warp as per-row shifts: tests, then the bench both ways
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "stereo_homography or trainer_mono or homography" > gpurun_out/r2/pytest_stereo.log 2>&1; echo "pytest rc=$?"
grep -E "^E  +(Assertion|assert [0-9])|^FAILED|passed|failed|Error" gpurun_out/r2/pytest_stereo.log | cut -c1-300 | head -30
for f in "" "--general_stereo"; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --warp_type homography_warp $f > gpurun_out/r2/bench_stereo$f.log 2>&1
  echo "bench $f rc=$? $(grep -o '"value": [0-9.]*' gpurun_out/r2/bench_stereo$f.log | head -1) $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/r2/bench_stereo$f.log | head -1)"
done
tail -3 gpurun_out/r2/bench_stereo.log | cut -c1-600
