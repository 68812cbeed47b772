# Round 5, second GPU session: row-mode parity table (exact / thresholds / fast), workgroup shapes x shift ring x fixed-reference
# softmax under the three row modes, the backward's shift prefetch.
mkdir -p gpurun_out/r5
O=gpurun_out/r5
AB="python scripts/diag_kernel_ab.py --rounds 5 --iters 40"
V="s5 s4 s4f r1s5 r1s4 r1s5f r1s4f r2s5 r2s5f bshpf"
echo "== exact rows" | tee $O/ab2.txt
$AB --impl 6 product $V --out $O/ab2_exact.json 2>&1 | grep -v amdgpu.ids | tee -a $O/ab2.txt
echo "== second rows below 4e-6 dropped (PD_ROW_EPS)" | tee -a $O/ab2.txt
PD_ROW_EPS=4e-6 $AB --impl 0 product $V --out $O/ab2_eps4.json 2>&1 | grep -v amdgpu.ids | tee -a $O/ab2.txt
echo "== fast rows" | tee -a $O/ab2.txt
$AB --impl 2 product $V --out $O/ab2_fast.json 2>&1 | grep -v amdgpu.ids | tee -a $O/ab2.txt
echo "== parity of the variants' forward" | tee $O/var_tests2.txt
for v in s4f r1s5f r2s5; do
  PD_LIB=planedepth_amd/lib/libpd_var_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
    -k "segment_stream_forward or (fullsize_known and headline) or render_probability_on_the_row or degenerate" 2>&1 | tail -3 | sed "s/^/$v: /" | tee -a $O/var_tests2.txt
done
echo "== parity report" | tee $O/parity.log
timeout 2400 python scripts/parity_report.py --eps 2.9e-6,4e-6 --out $O/r05_parity_rows.md --no_part2 2>&1 | tail -30 | tee -a $O/parity.log
