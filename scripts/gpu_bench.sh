# quick perf check: bench only (extra bench flags after --)
mkdir -p gpurun_out
timeout 600 python bench.py --steps 30 --warmup 5 --no_cpu_baseline "$@" > gpurun_out/bench_quick.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench_quick.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels'))" || tail -5 gpurun_out/bench_quick.log
