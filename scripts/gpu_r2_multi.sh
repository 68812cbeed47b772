mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "views_as_one or trainer_mono or fixture or smooth or post_process" > gpurun_out/r2/pytest_multi.log 2>&1; echo "pytest rc=$?"
grep -E "^E  +(Assertion|assert [0-9])|^FAILED|passed|failed|Error" gpurun_out/r2/pytest_multi.log | cut -c1-300 | head
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_next_rows --no_ddp_step "$@" > gpurun_out/r2/m_$name.log 2>&1; echo "$name rc=$? $(grep -o '"value": [0-9.]*' gpurun_out/r2/m_$name.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2/m_$name.log | head -1) $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/r2/m_$name.log | head -1)"; }
b sides --warp_type homography_warp --mono_sides
b sides63 --warp_type homography_warp --mono_sides --automask --xz_levels 14
b sides_pernode --warp_type homography_warp --mono_sides --per_view_nodes
b headline
grep -h -i "Traceback" -A 12 gpurun_out/r2/m_*.log | head -30
