# rocprofv3 kernel stats of the default bench including the "next rows" operators -> warp / tail / smooth kernels
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/prof_next
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_next -o k -- python $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline > $REPO/gpurun_out/prof_next/run.log 2>&1); echo "rocprof rc=$?"
grep -E "warp_|tail_|smooth_|cat_flip" gpurun_out/prof_next/k_kernel_stats.csv | cut -d, -f1-4 | sed 's/(pd::[A-Za-z]*Args[^"]*//' | cut -c1-110
