mkdir -p gpurun_out/r2
for v in product $VARIANTS; do
  for w in 1 2 3; do
    if [ $v = product ]; then unset PD_LIB; else export PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so; fi
    echo -n "$v waves=$w "; PD_QUAD_WAVES=$w timeout 200 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_next_rows 2>/dev/null | grep -o '"kernels": {[^}]*}'
  done
done
