# One table of the secondary configurations (images/s, kernel times where the direct-launch timing applies)
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows "$@" > gpurun_out/t_$name.log 2>&1; echo "| $name | \`$*\` | $(grep -o '"value": [0-9.]*' gpurun_out/t_$name.log | head -1 | cut -d' ' -f2) | $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/t_$name.log) |"; }
echo "| config | bench.py flags | images/s | kernels |"
echo "|---|---|---|---|"
b headline
b n63_xz_automask --planes 49 --xz_levels 14 --automask
b batch12 --batch 12
b hr_384x1280 --batch 4 --height 384 --width 1280
b l1_no_mixture --no_mixture
b homography_49 --warp_type homography_warp
b homography_63_automask --warp_type homography_warp --planes 49 --xz_levels 14 --automask
PD_SWEEP_IMPL=2 b fast_rows_optin
