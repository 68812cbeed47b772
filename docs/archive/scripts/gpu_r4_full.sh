# Round 4: the whole GPU suite + the default bench line (what the driver runs at round end).
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r4/pytest_full.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r4/pytest_full.log
timeout 900 python bench.py > gpurun_out/r4/bench_default.log 2> gpurun_out/r4/bench_default.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/r4/bench_default.log
