export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests -m gpu -q -k "tile_backward or homography or random_cases or degenerate or fixture_vs_reference or trainer_mono" > gpurun_out/r2/pytest_tile.log 2>&1; echo "pytest-tile rc=$?"
grep -E "^E  +(Assertion|assert [0-9])|^FAILED|passed|failed" gpurun_out/r2/pytest_tile.log | cut -c1-200 | head -30
timeout 600 python scripts/diag_homog.py > gpurun_out/r2/diag_homog.log 2>&1; echo "diag rc=$?"; cat gpurun_out/r2/diag_homog.log | tail -30
for cfg in "stereo:--warp_type homography_warp" "mono:--warp_type homography_warp --mono_pose"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r2/prof_tile_$name -o k -- python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows $flags > $REPO/gpurun_out/r2/prof_tile_$name.log 2>&1); echo "rocprof $name rc=$?"
  tail -1 gpurun_out/r2/prof_tile_$name.log | cut -c1-200
  head -8 gpurun_out/r2/prof_tile_$name/k_kernel_stats.csv | cut -d, -f1-5 | cut -c1-160
done
