# Round 4: parity subset (PYTEST_K) + A/B of the step with the persistent forward on / off (PD_FWD_PERSIST) and $VARIANTS.
mkdir -p gpurun_out/r4
timeout ${PYTEST_TIMEOUT:-1200} python -m pytest tests -m gpu -q -x ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/r4/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert " gpurun_out/r4/pytest.log | tail -15
b() { name=$1; shift; timeout 300 env "$@" python bench.py --steps ${STEPS:-50} --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step $CFG > gpurun_out/r4/ab_$name.log 2>&1; echo "$name $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*|"isolated_[a-z]*_ms": [0-9.]*' gpurun_out/r4/ab_$name.log | head -6 | tr '\n' ' ')"; tail -3 gpurun_out/r4/ab_$name.log | grep -iE "error|Traceback" ; }
for rep in $(seq 1 ${REP:-2}); do
  b stream_$rep PD_DUMMY=1
  :
  for v in $VARIANTS; do b ${v}_$rep PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so; done
done
