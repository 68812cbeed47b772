# Round 5: A/B of library variants on the headline kernels in one process (scripts/diag_kernel_ab.py), per row mode.
#   VARIANTS="name ..." (built beforehand by scripts/build_variants.sh; "product" is always the first arm)
#   IMPLS="6 2" (pd_sweep_impl values: 6 exact rows, 2 fast rows, 0 auto)   SHAPES="headline b12 hr"   TESTS="variant ..." (parity subset under PD_LIB)
mkdir -p gpurun_out/r5
O=gpurun_out/r5/${TAG:-ab}.txt; : > $O
for impl in ${IMPLS:-6}; do
  for shape in ${SHAPES:-headline}; do
    case $shape in b12) f="--batch 12";; hr) f="--batch 4 --height 384 --width 1280";; *) f="";; esac
    echo "== impl $impl, $shape" | tee -a $O
    timeout 400 python scripts/diag_kernel_ab.py --rounds 5 --iters 40 --impl $impl $f product $VARIANTS 2>&1 | grep -v amdgpu.ids | tee -a $O
  done
done
for v in $TESTS; do
  PD_LIB=planedepth_amd/lib/libpd_var_$v.so timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
    -k "${TEST_K:-segment_stream_forward or fullsize_known or degenerate or fixture_vs_reference or random_cases or render_probability_on_the_row or fused_mean}" 2>&1 | tail -3 | sed "s/^/$v: /" | tee -a $O
done
