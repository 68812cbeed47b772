mkdir -p gpurun_out/r2
timeout 1200 python -m pytest tests -m gpu -q -x -k "rowquad" > gpurun_out/r2/pytest_quad.log 2>&1; echo "pytest-quad rc=$?"
grep -E "^E  +(Assertion|assert [0-9])|^FAILED|passed|failed|Error" gpurun_out/r2/pytest_quad.log | cut -c1-300 | head -20
timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows > gpurun_out/r2/b_quad.log 2>&1; tail -1 gpurun_out/r2/b_quad.log | cut -c1-1200
PD_SWEEP_IMPL=4 timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows > gpurun_out/r2/b_rows1.log 2>&1; tail -1 gpurun_out/r2/b_rows1.log | cut -c1-400
