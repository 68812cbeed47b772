# Plane-uniform kernels: parity subset, then kernel times (--mono_pose under rocprofv3) and the --mono_sides step, product + $VARIANTS.
export TMPDIR=/tmp
REPO=$PWD
OUT=gpurun_out/r4/uni; mkdir -p $OUT
if [ -z "$NO_TESTS" ]; then
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "uniform or homography or shortcut or mono or trainer" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
fi
for v in prod $VARIANTS; do
  export PD_PAIR_FORWARD=1; if [ $v = nopair ]; then export PD_PAIR_FORWARD=0; fi
  if [ $v != prod ] && [ $v != nopair ]; then export PD_LIB=$REPO/planedepth_amd/lib/libpd_var_$v.so; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/$v -o k -- python $REPO/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --warp_type homography_warp ${PROF_CFG:---mono_sides} --launch eager > $REPO/$OUT/$v.log 2>&1); echo "$v rc=$?"
  python - $OUT/$v/k_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'uniform_' in r['Name'] and float(r['AverageNs']) > 20000: print('   %-50s %4s  %9.1f us' % (r['Name'].split('(')[0][-50:], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  for rep in 1 2; do
    timeout 400 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --warp_type homography_warp --mono_sides > $OUT/sides_${v}_$rep.log 2>&1
    echo "sides $v $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"launch": "[a-z]*"' $OUT/sides_${v}_$rep.log | head -3 | tr '\n' ' ')"
  done
done
