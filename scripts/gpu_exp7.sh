export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/profh
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/profh -o homo -- python $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_next_rows --warp_type homography_warp --planes 49 --xz_levels 14 --automask > $REPO/gpurun_out/profh_run.log 2>&1); echo "rocprof rc=$?"
tail -1 gpurun_out/profh_run.log | cut -c1-200
