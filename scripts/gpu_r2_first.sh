# round 2, first session: new parity tests + kernel stats of the general (homography) kernels before the rewrite
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests -m gpu -q -k "trainer_mono or fullsize_63 or cpu_batch or spawns" > gpurun_out/r2/pytest_new.log 2>&1; echo "pytest-new rc=$?"
grep -E "^E  |FAILED|passed|failed" gpurun_out/r2/pytest_new.log | head -30
for cfg in "stereo:--warp_type homography_warp" "mono:--warp_type homography_warp --mono_pose"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r2/prof_homog_$name -o k -- python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows $flags > $REPO/gpurun_out/r2/prof_homog_$name.log 2>&1); echo "rocprof $name rc=$?"
  tail -1 gpurun_out/r2/prof_homog_$name.log | cut -c1-300
  head -8 gpurun_out/r2/prof_homog_$name/k_kernel_stats.csv | cut -d, -f1-5 | cut -c1-160
done
