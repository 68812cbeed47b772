# Round 4: BASELINE configs[3] as the trainer runs it (--mono_sides) and one novel frame alone (--mono_pose), product + $VARIANTS.
mkdir -p gpurun_out/r4
b() { name=$1; shift; timeout 400 env "$@" python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --warp_type homography_warp $CFG > gpurun_out/r4/s_$name.log 2>&1; echo "$name $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/r4/s_$name.log | head -4 | tr '\n' ' ')"; tail -3 gpurun_out/r4/s_$name.log | grep -iE "error|Traceback"; }
for rep in $(seq 1 ${REP:-2}); do
  CFG=--mono_sides b sides_$rep PD_DUMMY=1
  CFG=--mono_pose b mono_$rep PD_DUMMY=1
  for v in $VARIANTS; do CFG=--mono_sides b sides_${v}_$rep PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so; CFG=--mono_pose b mono_${v}_$rep PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so; done
done
