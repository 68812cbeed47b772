# A/B harness: persistent row grid vs one workgroup per row, and other environment-selectable variants
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows "$@" > gpurun_out/ab_$name.log 2>&1; echo "$name $(grep -o '"value": [0-9.]*' gpurun_out/ab_$name.log | head -1) $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/ab_$name.log)"; }
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E   |FAILED|passed|failed" | head
b persistent
PD_ROW_GRID=0 b classic
PD_ROW_GRID=512 b g512
PD_ROW_GRID=1024 b g1024
b persistent2
PD_ROW_GRID=0 b classic2
