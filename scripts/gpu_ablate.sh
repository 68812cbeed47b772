mkdir -p gpurun_out
for ab in 0 4 6 12 14; do
  if [ $ab = 0 ]; then unset PD_LIB; else export PD_LIB=$PWD/planedepth_amd/lib/libpd_ablate_$ab.so; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline > gpurun_out/ab_$ab.log 2>&1
  echo -n "ablate=$ab "; tail -1 gpurun_out/ab_$ab.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('kernels'))"
done
