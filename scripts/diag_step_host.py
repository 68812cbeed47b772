"""Host cost of the eager headline step (bench.py's own step): issue time per step back to back, then cProfile of 1000 steps (the backward's Python runs on
autograd's device thread: it shows up inside run_backward only).  python scripts/diag_step_host.py"""
import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py"]
import torch, bench
args = bench.parse()
dev = torch.device("cuda:0")
import __graft_entry__ as e; e.build()
c = bench.make_batch(args, dev, seed=0)
step, _ = bench.build_step(args, c, dev)
if os.environ.get('PD_SINGLE_THREAD_AUTOGRAD') == '1':
    torch.autograd.set_multithreading_enabled(False)   # the backward's nodes run on the calling thread: no hand-over to autograd's device thread
for _ in range(200): step()
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze(); gc.disable()
# host-only cost: time per step while the device queue is kept short (sync every step) vs back to back
t0 = time.perf_counter()
for _ in range(1000): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("back to back: %.1f us per step issued (the launch queue fills: this tends to the device's rate), %.1f us incl. drain" % ((t1 - t0) / 1000 * 1e6, (t2 - t0) / 1000 * 1e6))
issued = 0.0
for _ in range(100):   # the host's own cost: ten steps at a time into an empty queue
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    issued += time.perf_counter() - t0
torch.cuda.synchronize()
print("host alone: %.1f us per step (10 steps into an empty queue, 100 times)" % (issued / 1000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(1000): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
