# Functional check of the multi-process bench flow on a one-GPU box: 2 ranks share cuda:0, gloo for the barrier.
mkdir -p gpurun_out
PD_BENCH_BACKEND=gloo PD_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/bench_mp.log 2>&1; echo "rc=$?"
tail -1 gpurun_out/bench_mp.log | cut -c1-400
