# A/B of library variants (scripts/build_variants.sh) inside the training step: $VARIANTS names, product first and last.
mkdir -p gpurun_out
b() { name=$1; shift; timeout 300 python bench.py --steps ${STEPS:-30} --warmup 5 --no_cpu_baseline --no_next_rows $CFG "$@" > gpurun_out/v_$name.log 2>&1; echo "$name $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"frac": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*|"isolated_[a-z]*_ms": [0-9.]*' gpurun_out/v_$name.log | head -7 | tr '\n' ' ')"; }
b product
for v in $VARIANTS; do PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so b $v; done
b product2
if [ -n "$QUAD" ]; then PD_QUAD_FWD=1 b quadfwd; fi
if [ -n "$QUADV" ]; then for v in $QUADV; do PD_QUAD_FWD=1 PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so b quadfwd_$v; done; fi
