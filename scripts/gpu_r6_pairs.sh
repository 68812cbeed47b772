# Round 6, NOTEBOOK 11.1: the row-pair backward (experiments library, PD_BWD_PAIRS=1) against the product kernel: parity test + A/B.
mkdir -p gpurun_out/r6
bash scripts/build_experiments.sh 2>&1 | tail -1
O=gpurun_out/r6/pairs_exp.txt
PD_TEST_EXPERIMENTS=1 PD_LIB=$PWD/planedepth_amd/lib/libpd_experiments.so timeout 900 python -m pytest tests/experiments -q -k row_pair 2>&1 | tail -4 | tee $O
for f in "" "--batch 12"; do
  PD_BWD_PAIRS=1 timeout 300 python scripts/diag_kernel_ab.py --check --rounds 5 --iters 40 $f product pairs=planedepth_amd/lib/libpd_experiments.so 2>&1 | grep -v amdgpu.ids | tee -a $O
done
