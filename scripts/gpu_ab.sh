# A/B harness: environment-selectable variants of the default workload
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E   |FAILED|passed|failed" | head
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows "$@" > gpurun_out/ab_$name.log 2>&1; echo "$name $(grep -o '"value": [0-9.]*' gpurun_out/ab_$name.log | head -1) $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/ab_$name.log)"; }
b pairs
PD_NO_ROWPAIR=1 b nopairs
b pairs2
PD_NO_ROWPAIR=1 b nopairs2
