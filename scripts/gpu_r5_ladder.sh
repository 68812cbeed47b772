# Round 5, VERDICT r4 #1: the segment-stream forward's ablation ladder, the per-workgroup timeline, A/B of the candidate
# changes, and the parity suite under both row modes.  Variants come from scripts/build_variants.sh (built before the call).
mkdir -p gpurun_out/r5
O=gpurun_out/r5
AB="python scripts/diag_kernel_ab.py --rounds 6 --iters 40"
echo "== A/B, exact rows" | tee $O/ab.txt
$AB --impl 6 r4 base occ4 shring shring4 colpf4 --out $O/ab_exact.json 2>&1 | tee -a $O/ab.txt
echo "== A/B, fast rows" | tee -a $O/ab.txt
$AB --impl 2 r4 base occ4 shring shring4 colpf4 --out $O/ab_fast.json 2>&1 | tee -a $O/ab.txt
echo "== ladder, exact rows" | tee $O/ladder.txt
$AB --impl 6 occ4 abl16 abl2 abl4 abl8 abl32 abl64 abl128 abl256 abl36 abl166 abl190 abl66 abl82 --out $O/ladder_exact.json 2>&1 | tee -a $O/ladder.txt
echo "== ladder, fast rows" | tee -a $O/ladder.txt
$AB --impl 2 occ4 abl2 abl4 abl8 abl32 abl64 abl128 abl256 abl36 abl166 abl66 --out $O/ladder_fast.json 2>&1 | tee -a $O/ladder.txt
echo "== trace, exact rows" | tee $O/trace.txt
python scripts/diag_fwd_trace.py --impl 6 --out $O/fwd_trace_exact 2>&1 | tee -a $O/trace.txt
echo "== trace, fast rows" | tee -a $O/trace.txt
python scripts/diag_fwd_trace.py --impl 2 --out $O/fwd_trace_fast 2>&1 | tee -a $O/trace.txt
echo "== variants: parity of the forward" | tee $O/var_tests.txt
for v in shring4 colpf4 occ4; do
  PD_LIB=planedepth_amd/lib/libpd_var_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
    -k "segment_stream_forward or (fullsize_known and headline) or render_probability_on_the_row" 2>&1 | tail -3 | sed "s/^/$v: /" | tee -a $O/var_tests.txt
done
echo "== parity suite under both row modes" | tee $O/rowmode_tests.txt
timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -rf \
  -k "fixture_vs_reference or random_cases_vs or fullsize_known or fullsize_vs_oracle or other_baseline or degenerate_shapes or dense_disparity or band_limited or fast_rows" 2>&1 | tail -60 | tee -a $O/rowmode_tests.txt
