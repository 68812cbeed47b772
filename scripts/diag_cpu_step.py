"""Host-side cost of one step (cProfile over 300 steps; the device queue is drained every 50 so that nothing blocks)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = ["bench.py", "--no_cpu_baseline", "--no_next_rows", "--no_ddp_step"] + sys.argv[1:]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
import __graft_entry__ as e; e.build()
c = bench.make_batch(args, dev, 0)
step, _ = bench.build_step(args, c, dev)
for _ in range(60):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    step()
t_enq = (time.perf_counter() - t0) / 100
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 100
print("host enqueue %.3f ms per step; with the device %.3f ms per step" % (t_enq * 1e3, t_all * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(300):
    step()
    if i % 50 == 49:
        torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime"); st.print_stats(22)
