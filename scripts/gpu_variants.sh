mkdir -p gpurun_out
for v in $VARIANTS; do
  export PD_LIB=$PWD/planedepth_amd/lib/libpd_var_$v.so
  timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline $EXTRA > gpurun_out/var_$v.log 2>&1
  echo -n "variant=$v "; tail -1 gpurun_out/var_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('kernels'))"
done
