# round-2 check: smoke, the whole GPU suite, the default bench (with the ddp_step block), a 2-rank run on one device
mkdir -p gpurun_out/r2
python __graft_entry__.py --smoke > gpurun_out/r2/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r2/pytest_full.log 2>&1; echo "pytest-full rc=$?"
grep -E "^E  +(Assertion|assert [0-9])|^FAILED|passed|failed|Error" gpurun_out/r2/pytest_full.log | cut -c1-250 | head -30
timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/r2/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r2/bench.log | cut -c1-3000
PD_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 4 --no_cpu_baseline --no_next_rows > gpurun_out/r2/bench_2rank.log 2>&1; echo "bench2 rc=$?"; tail -1 gpurun_out/r2/bench_2rank.log | cut -c1-2500
