# A/B harness: run the default workload under environment-selectable variants on the same box, e.g.
#   PD_NO_ROWPAIR=1 (single-row forward), PD_SWEEP_IMPL=2 (fast rows), PD_LIB=<variant .so from scripts/build_variants.sh>
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows "$@" > gpurun_out/ab_$name.log 2>&1; echo "$name $(grep -o '"value": [0-9.]*' gpurun_out/ab_$name.log | head -1) $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/ab_$name.log)"; }
b default
PD_NO_ROWPAIR=1 b nopairs
PD_SWEEP_IMPL=2 b fast_rows
b default2
