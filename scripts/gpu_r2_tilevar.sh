# tile backward: ablation variants + the wide-access probe
mkdir -p gpurun_out/r2
b() { name=$1; shift; timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_next_rows --warp_type homography_warp "$@" > gpurun_out/r2/tv_$name.log 2>&1; echo "$name $(grep -oE '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/r2/tv_$name.log | head -1) $(grep -oE '"ms_per_step": [0-9.]*' gpurun_out/r2/tv_$name.log)"; }
for pose in "" "--mono_pose" "--colmap_pose"; do
  echo "pose: $pose"
  b product $pose
  PD_SWEEP_IMPL=3 b atomic $pose
  for v in $VARIANTS; do PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so b $v $pose; done
done
./scripts/probes/wide_probe > gpurun_out/r2/wide_probe.log 2>&1; cat gpurun_out/r2/wide_probe.log
