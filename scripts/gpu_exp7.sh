export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/prof63
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof63 -o n63 -- python $REPO/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows --planes 49 --xz_levels 14 --automask > $REPO/gpurun_out/prof63_run.log 2>&1); echo "rocprof rc=$?"
cut -c1-120 gpurun_out/prof63/n63_kernel_stats.csv | head -16
