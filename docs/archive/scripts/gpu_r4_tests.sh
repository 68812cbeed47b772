# Round 4: a pytest selection on the product library, then (EXP=1) the experiments library's own tests.
mkdir -p gpurun_out/r4
timeout ${PYTEST_TIMEOUT:-2400} python -m pytest tests -m gpu -q -x ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/r4/pytest_sel.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert " gpurun_out/r4/pytest_sel.log | tail -12
if [ -n "$EXP" ]; then
  PD_TEST_EXPERIMENTS=1 PD_LIB=$PWD/planedepth_amd/lib/libpd_experiments.so timeout 1800 python -m pytest tests/experiments -q -x > gpurun_out/r4/pytest_exp.log 2>&1; echo "experiments rc=$?"
  tail -3 gpurun_out/r4/pytest_exp.log
fi
