# Per-kernel A/B of library variants under rocprofv3: VARIANTS="prev" CFG="..." bash scripts/gpu_prof_ab.sh
export TMPDIR=/tmp
REPO=$PWD
for v in product $VARIANTS; do
  if [ $v != product ]; then export PD_LIB=$REPO/planedepth_amd/lib/libpd_var_$v.so; fi
  mkdir -p gpurun_out/ab_$v
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/ab_$v -o k -- python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows $CFG > $REPO/gpurun_out/ab_$v/run.log 2>&1)
  echo "== $v"; head -4 gpurun_out/ab_$v/k_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
