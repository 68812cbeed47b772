# Round 6: the whole GPU suite on the current build, then the default bench line.
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r6/smoke.log 2>&1; echo "smoke rc=$?"
timeout 3000 python -m pytest tests/ -q -m gpu ${PYTEST_ARGS:-} > gpurun_out/r6/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r6/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r6/bench.json 2> gpurun_out/r6/bench.err; echo "bench rc=$?"; python scripts/show_bench.py gpurun_out/r6/bench.json 2>/dev/null | head -20
