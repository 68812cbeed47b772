# Round 4: eager step vs HIP-graph replay of the captured step, alternating, on one box (host-paced or not?)
mkdir -p gpurun_out/r4
b() { name=$1; shift; timeout 300 python bench.py --steps ${STEPS:-50} --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step "$@" > gpurun_out/r4/g_$name.log 2>&1; echo "$name $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/r4/g_$name.log | head -4 | tr '\n' ' ')"; }
for rep in 1 2 3; do
  b eager_$rep
  b graph_$rep --hip_graph
  STEPS=20 b eager20_$rep
  STEPS=20 b graph20_$rep --hip_graph
done
python scripts/diag_host_overhead.py 2>&1 | tail -5
