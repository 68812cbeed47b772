# Round 5: headline profile set, configuration table (+ configs[3] as specified with stats and traffic), parity report.
mkdir -p gpurun_out/r5
bash scripts/gpu_r5_profile.sh > gpurun_out/r5/profile.log 2>&1; tail -4 gpurun_out/r5/profile.log
bash scripts/gpu_r5_configs.sh > gpurun_out/r5/configs.log 2>&1; tail -25 gpurun_out/r5/configs.log
timeout 2400 python scripts/parity_report.py --eps 2.9e-6,4e-6 --out gpurun_out/r5/r05_parity.md > gpurun_out/r5/parity_full.log 2>&1; tail -3 gpurun_out/r5/parity_full.log
