# The reference's default plane set (49 xy + 14 xz planes, horizon mask, automask): parity subset + the bench line + kernel stats.
export TMPDIR=/tmp
REPO=$PWD
OUT=gpurun_out/r4/xz; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "xz or n63 or fixture or trainer or fullsize or rowshift or stream" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for rep in 1 2 3; do
  timeout 400 python bench.py --steps 50 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --xz_levels 14 --automask > $OUT/b_$rep.log 2>&1
  echo "n63_xz $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*|"launch": "[A-Za-z ]*' $OUT/b_$rep.log | head -4 | tr '\n' ' ')"
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o k -- python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --xz_levels 14 --automask --launch eager > $REPO/$OUT/prof.log 2>&1); echo "prof rc=$?"
head -8 $OUT/prof/k_kernel_stats.csv | cut -c1-150
