# stereo side of homography_warp as per-row shifts + the three-view step: tests, then the benches
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "stereo_homography or trainer_mono or homography or views_as_one or uniform" > gpurun_out/r2/pytest_stereo.log 2>&1; echo "pytest rc=$?"
grep -E "^E  +(Assertion|assert [0-9])|^FAILED|passed|failed|Error" gpurun_out/r2/pytest_stereo.log | cut -c1-300 | head -30
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step "$@" > gpurun_out/r2/st_$name.log 2>&1; echo "$name rc=$? $(grep -o '"value": [0-9.]*' gpurun_out/r2/st_$name.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2/st_$name.log | head -1) $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/r2/st_$name.log | head -1)"; }
b stereo --warp_type homography_warp
b stereo_general --warp_type homography_warp --general_stereo
b sides --warp_type homography_warp --mono_sides
b sides_per_view --warp_type homography_warp --mono_sides --per_view_nodes
b mono --warp_type homography_warp --mono_pose
