# Round 3 loop: GPU parity suite (optionally -k $PYTEST_K) + headline bench with the row-stream backward and, for A/B on
# the same box, with the row-shift backward (PD_SWEEP_IMPL=4 = PD_IMPL_ROWS1).
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/pytest_r3.log 2>&1; echo "pytest rc=$?"
tail -${TAIL:-25} gpurun_out/pytest_r3.log
b() { name=$1; shift; timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows $CFG "$@" > gpurun_out/r3_$name.log 2>&1; echo "$name $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"frac": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*|"isolated_[a-z]*_ms": [0-9.]*' gpurun_out/r3_$name.log | head -8 | tr '\n' ' ')"; }
b stream
b stream2
PD_SWEEP_IMPL=4 b shift
