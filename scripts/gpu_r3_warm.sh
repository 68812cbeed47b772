# How long does the slow stretch at the start of a process last?  Same 20 timed steps after 5 / 20 / 50 / 200 warm-up steps.
mkdir -p gpurun_out
b() { name=$1; shift; timeout 300 python bench.py --steps 20 --no_cpu_baseline --no_next_rows "$@" > gpurun_out/w_$name.log 2>&1; echo "$name $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/w_$name.log | head -3 | tr '\n' ' ')"; }
for w in 5 20 50 200 5 200; do b warm$w --warmup $w; done
