# parity + headline bench + the secondary configs (N=63 with mask/automask rows path, HR)
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline "$@" > gpurun_out/b_$name.log 2>&1; echo "$name $(grep -o '"value": [0-9.]*' gpurun_out/b_$name.log | head -1) $(grep -o '"fwd_ms": [0-9.]*, "bwd_ms": [0-9.]*' gpurun_out/b_$name.log)"; }
b head
b w512 --width 512
b w768 --width 768
b n63 --planes 49 --xz_levels 14 --automask
b hr --batch 4 --height 384 --width 1280
b b12 --batch 12
