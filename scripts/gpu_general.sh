# general-kernel configurations: parity + homography / mono benches
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E   |FAILED|passed|failed" | head -20
b() { name=$1; shift; timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_next_rows "$@" > gpurun_out/b_$name.log 2>&1; echo "$name $(grep -o '"value": [0-9.]*' gpurun_out/b_$name.log | head -1)"; }
b homo63 --warp_type homography_warp --planes 49 --xz_levels 14 --automask
b homo49 --warp_type homography_warp
