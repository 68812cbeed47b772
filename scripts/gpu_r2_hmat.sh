mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "homography or trainer_mono or uniform or fixture" > gpurun_out/r2/pytest_hmat.log 2>&1; echo "pytest rc=$?"
grep -E "^E  +(Assertion|assert [0-9])|^FAILED|passed|failed|Error" gpurun_out/r2/pytest_hmat.log | cut -c1-300 | head -30
bash scripts/gpu_r2_graph.sh
