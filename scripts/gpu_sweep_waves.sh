mkdir -p gpurun_out
for w in 1 2 4 5 8 10; do
  PD_ROW_WAVES=$w timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/bw_$w.log 2>&1
  echo -n "waves=$w "; tail -1 gpurun_out/bw_$w.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels'))"
done
