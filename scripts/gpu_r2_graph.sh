# HIP-graph replay of the step vs eager launches, per configuration
mkdir -p gpurun_out/r2
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step "$@" > gpurun_out/r2/g_$name.log 2>&1; echo "$name rc=$? $(grep -o '"value": [0-9.]*' gpurun_out/r2/g_$name.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2/g_$name.log | head -1) $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/r2/g_$name.log | head -1)"; }
for g in "" "--hip_graph"; do
b headline$g $g
b homo_stereo$g --warp_type homography_warp $g
b homo_mono$g --warp_type homography_warp --mono_pose $g
b homo_colmap$g --warp_type homography_warp --colmap_pose $g
b n63$g --planes 49 --xz_levels 14 --automask $g
done
grep -h -i "error\|Traceback" gpurun_out/r2/g_*.log | head
