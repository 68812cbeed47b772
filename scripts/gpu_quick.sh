# Quick GPU loop: sweep parity tests (subset by -k) + A/B bench of the product library against variants in $VARIANTS.
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert " gpurun_out/pytest_quick.log | tail -15
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows $CFG "$@" > gpurun_out/q_$name.log 2>&1; echo "$name $(grep -o '"value": [0-9.]*' gpurun_out/q_$name.log | head -1) $(grep -oE '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/q_$name.log | head -1) $(grep -oE '"ms_per_step": [0-9.]*' gpurun_out/q_$name.log)"; }
b product
for v in $VARIANTS; do PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so b $v; done
b product2
