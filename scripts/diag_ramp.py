"""Per-step GPU time of the first steps of a fresh process (is the slow start allocator growth or clock ramp?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = ["bench.py", "--no_cpu_baseline", "--no_next_rows", "--no_ddp_step"]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
import __graft_entry__ as e; e.build()
c = bench.make_batch(args, dev, 0)
step, _ = bench.build_step(args, c, dev)
if os.environ.get("PREWARM"):   # busy the device first: does the slow stretch depend on time-since-idle or on step count?
    x = torch.empty(64 << 20, device=dev)
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < float(os.environ["PREWARM"]):
        x.add_(1.0)
    torch.cuda.synchronize()
ev = []
t0 = time.perf_counter()
for i in range(80):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); step(); b.record(); ev.append((a, b))
torch.cuda.synchronize()
wall = time.perf_counter() - t0
ms = [a.elapsed_time(b) for a, b in ev]
print("wall %.1f ms for 80 steps" % (wall * 1e3))
print("steps 0-9  :", " ".join("%.3f" % m for m in ms[:10]))
print("steps 10-29:", " ".join("%.3f" % m for m in ms[10:30]))
print("steps 60-79:", " ".join("%.3f" % m for m in ms[60:]))
print("alloc retries", torch.cuda.memory_stats()["num_alloc_retries"], "reserved MB", torch.cuda.memory_reserved() // 2**20)
