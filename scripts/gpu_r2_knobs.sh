# In-step A/B of the row kernels' compile-time knobs (variants from scripts/build_variants.sh), same box, interleaved twice
mkdir -p gpurun_out/r2
b() { timeout 300 python bench.py --steps 100 --warmup 10 --no_cpu_baseline --no_next_rows --no_ddp_step 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*\|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' | tr "\n" " "; }
for round in 1 2; do
  for v in product bu4 bu1 fu2 fu8 bocc2 bocc4 focc2 focc4; do
    if [ $v = product ]; then unset PD_LIB; else export PD_LIB=planedepth_amd/lib/libpd_var_$v.so; fi
    echo "$round $v $(b)"
  done
done
