cd $GRAFT_REPO_ROOT 2>/dev/null || true
b() { name=$1; shift; timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows "$@" > gpurun_out/g_$name.log 2>&1; echo "$name $(grep -oE '"value": [0-9.]*|"ms_per_step": [0-9.]*|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/g_$name.log | head -3 | tr '\n' ' ')"; }
mkdir -p gpurun_out
b eager1; b graph1 --hip_graph; b eager2; b graph2 --hip_graph; b eager100 --steps 100 --warmup 10; b graph100 --hip_graph --steps 100 --warmup 10
