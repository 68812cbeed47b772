# rocprofv3 kernel stats of one bench configuration: CFG="--warp_type homography_warp" NAME=homog bash scripts/gpu_prof_cfg.sh
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/prof_$NAME
python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows $CFG 2>&1 | tail -1 | cut -c1-400
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$NAME -o k -- python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows $CFG > $REPO/gpurun_out/prof_$NAME/run.log 2>&1); echo "rocprof rc=$?"
head -25 gpurun_out/prof_$NAME/k_kernel_stats.csv | cut -d, -f1-5 | cut -c1-200
