mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2/pytest_full.log 2>&1; echo "pytest-full rc=$?"
grep -E "^E  +(Assertion|assert [0-9])|^FAILED|passed|failed" gpurun_out/r2/pytest_full.log | cut -c1-200 | head -30
timeout 600 python scripts/diag_homog.py > gpurun_out/r2/diag_homog.log 2>&1; echo "diag rc=$?"; grep -v Warning gpurun_out/r2/diag_homog.log | tail -16
