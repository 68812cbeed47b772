# kernel-level times per variant (rocprofv3 stats)
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/vp
for v in $VARIANTS; do
  if [ $v = base ]; then unset PD_LIB; else export PD_LIB=$REPO/planedepth_amd/lib/libpd_var_$v.so; fi
  rm -rf gpurun_out/vp/$v
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/vp/$v -o s -- python $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline $EXTRA > $REPO/gpurun_out/vp/$v.log 2>&1)
  echo "== $v"; grep -E "rowstage|rowshift" gpurun_out/vp/$v/s_kernel_stats.csv | awk -F'","' '{printf "   %-70s avg %8.1f us\n", substr($1,2,70), $4/1000}'
done
