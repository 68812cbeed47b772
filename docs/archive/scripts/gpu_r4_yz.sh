# Dense disparity maps (49 xy + 14 yz planes: the general forward, the row-dense backward) against the atomic backward (PD_SWEEP_IMPL=1).
export TMPDIR=/tmp
REPO=$PWD
OUT=gpurun_out/r4/yz; mkdir -p $OUT
for impl in 0 1; do
  for rep in 1 2; do
    PD_SWEEP_IMPL=$impl timeout 400 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --yz_levels 14 --automask > $OUT/b_${impl}_$rep.log 2>&1
    echo "impl=$impl $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' $OUT/b_${impl}_$rep.log | head -3 | tr '\n' ' ')"; tail -2 $OUT/b_${impl}_$rep.log | grep -iE "error|Traceback"
  done
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --yz_levels 14 --automask --launch eager > $REPO/$OUT/prof.log 2>&1); echo "prof rc=$?"
head -6 $OUT/prof/k_kernel_stats.csv | cut -c1-140
