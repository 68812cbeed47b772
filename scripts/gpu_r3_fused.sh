# A/B of the fused plane-uniform backward variants (experiments builds) in the --mono_pose step: $VARIANTS
mkdir -p gpurun_out
b() { name=$1; shift; timeout 300 python bench.py --steps ${STEPS:-30} --warmup 5 --no_cpu_baseline --no_next_rows --warp_type homography_warp --mono_pose "$@" > gpurun_out/f_$name.log 2>gpurun_out/f_$name.err; echo "$name $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/f_$name.log | head -4 | tr '\n' ' ') $(tail -1 gpurun_out/f_$name.err | cut -c1-200)"; }
b product
for v in $VARIANTS; do PD_UNI_FUSED=1 PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so b $v; done
b product2
