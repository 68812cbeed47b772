# A/B: row pairs on/off x exact/fast rows (forward/backward kernel times of the default workload)
b() { name=$1; shift; timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows "$@" > gpurun_out/ab_$name.log 2>&1; echo "$name $(grep -o '"value": [0-9.]*' gpurun_out/ab_$name.log | head -1) $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/ab_$name.log)"; }
b pair_exact
PD_SWEEP_IMPL=2 b pair_fast
PD_NO_ROWPAIR=1 b single_exact
PD_NO_ROWPAIR=1 PD_SWEEP_IMPL=2 b single_fast
