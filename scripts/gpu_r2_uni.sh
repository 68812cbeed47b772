mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests -m gpu -q -k "uniform or trainer_mono or homography" > gpurun_out/r2/pytest_uni.log 2>&1; echo "pytest-uni rc=$?"
grep -E "^E  +(Assertion|assert [0-9])|^FAILED|passed|failed|Error" gpurun_out/r2/pytest_uni.log | cut -c1-250 | head -20
for pose in "" "--mono_pose" "--colmap_pose"; do
  echo -n "homography $pose: "; timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_next_rows --warp_type homography_warp $pose 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('kernels'))"
done
