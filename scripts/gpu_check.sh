# One GPU-box session: smoke, GPU parity tests, bench, rocprofv3 kernel trace.  Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert " gpurun_out/pytest_gpu.log | tail -30
timeout 900 python scripts/diag_errors.py > gpurun_out/diag.log 2>&1; echo "diag rc=$?"; cat gpurun_out/diag.log | tail -32
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -2 gpurun_out/bench.log
export TMPDIR=/tmp
REPO=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o sweep -- python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline > $REPO/gpurun_out/prof_run.log 2>&1); echo "rocprof rc=$?"
ls -R gpurun_out/prof | head -20
