# A/B on the headline and the 384x1280 configuration: VARIANTS="prev" bash scripts/gpu_ab_hr.sh
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert " gpurun_out/pytest_quick.log | tail -8
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows "$@" > gpurun_out/q_$name.log 2>&1; echo "$name $(grep -o '"value": [0-9.]*' gpurun_out/q_$name.log | head -1) $(grep -oE '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/q_$name.log | head -1)"; }
for rep in 1 2; do
  b product
  for v in $VARIANTS; do PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so b $v; done
  b product_hr --batch 4 --height 384 --width 1280
  for v in $VARIANTS; do PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so b ${v}_hr --batch 4 --height 384 --width 1280; done
done
