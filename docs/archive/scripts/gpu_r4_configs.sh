# Round 4: the BASELINE configurations on the current build (bench.py flags as in profiles/r03_configs.md) + render_probability.
mkdir -p gpurun_out/r4
b() { name=$1; shift; timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step "$@" > gpurun_out/r4/c_$name.log 2>&1; echo "$name $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"frac": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/r4/c_$name.log | head -5 | tr '\n' ' ')"; }
b headline
b n63 --xz_levels 14 --automask
b b12 --batch 12
b hr --height 384 --width 1280 --batch 4
b l1 --no_mixture
b homo_stereo --warp_type homography_warp
b homo_mono --warp_type homography_warp --mono_pose
b homo_sides --warp_type homography_warp --mono_sides
b homo_colmap --warp_type homography_warp --colmap_pose
b render --render_probability
