mkdir -p gpurun_out/r2
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step "$@" > gpurun_out/r2/s_$name.log 2>&1; echo "$name rc=$? $(grep -o '"value": [0-9.]*' gpurun_out/r2/s_$name.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2/s_$name.log | head -1) $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/r2/s_$name.log | head -1)"; }
b sides --warp_type homography_warp --mono_sides
b sides_automask --warp_type homography_warp --mono_sides --automask
b sides63 --warp_type homography_warp --mono_sides --automask --xz_levels 14
PD_TORCH_HOMOGRAPHY=1 b sides_torchalgebra --warp_type homography_warp --mono_sides
b sides_general --warp_type homography_warp --mono_sides --general_stereo
grep -h -i "error\|Traceback" -A 8 gpurun_out/r2/s_*.log | head -30
