mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "uniform or trainer_mono" > gpurun_out/r2/pytest_uni.log 2>&1; echo "pytest rc=$?"
grep -E "^E  +(Assertion|assert [0-9])|^FAILED|passed|failed|Error" gpurun_out/r2/pytest_uni.log | cut -c1-300 | head
export TMPDIR=/tmp; REPO=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r2/uni2 -o k -- python $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --warp_type homography_warp --mono_pose > $REPO/gpurun_out/r2/uni2.log 2>&1)
head -6 gpurun_out/r2/uni2/k_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step --warp_type homography_warp --mono_pose 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*'
