# XCD-banded block order in the pixel-linear gather kernels: A/B against a -DPD_XCD_BAND=0 build
mkdir -p gpurun_out/r2
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step "$@" > gpurun_out/r2/band_$name.log 2>&1; echo "$name rc=$? $(grep -o '"value": [0-9.]*' gpurun_out/r2/band_$name.log | head -1) $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/r2/band_$name.log | head -1)"; }
for lib in "" planedepth_amd/lib/libpd_var_noband.so; do
  export PD_LIB=$lib; [ -z "$lib" ] && unset PD_LIB
  echo "== ${lib:-product (banded)}"
  b mono --warp_type homography_warp --mono_pose
  b colmap --warp_type homography_warp --colmap_pose
  b stereo_general --warp_type homography_warp --general_stereo
  b dense_disp --xz_levels 14
  PD_SWEEP_IMPL=1 b general_disp
done
unset PD_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "uniform or homography or general or render or fixture" 2>&1 | tail -2
