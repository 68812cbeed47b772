# Round 6: A/B of library variants / impls on the headline kernels in one process (scripts/diag_kernel_ab.py).
#   ARMS="product product@7 var1 ..." (name@impl: that library under another pd_sweep_impl; variants built by scripts/build_variants.sh)
#   SHAPES="headline b12 hr n63"   CHECK=1 (bitwise comparison of every arm with the first)   TEST_K="pytest -k expression" (parity subset)
mkdir -p gpurun_out/r6
O=gpurun_out/r6/${TAG:-ab}.txt; : > $O
for shape in ${SHAPES:-headline}; do
  case $shape in b12) f="--batch 12";; hr) f="--batch 4 --height 384 --width 1280";; n63) f="--planes 63 --automask";; *) f="";; esac
  echo "== $shape" | tee -a $O
  timeout 600 python scripts/diag_kernel_ab.py --rounds ${ROUNDS:-5} --iters 40 --impl ${IMPL:-0} ${CHECK:+--check} $f ${ARMS:-product product@7} 2>&1 | grep -v amdgpu.ids | tee -a $O
done
if [ -n "$TEST_K" ]; then
  timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "$TEST_K" 2>&1 | tail -5 | tee -a $O
fi
