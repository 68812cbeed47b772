"""Which torch / HIP kernels one bench step launches (torch.profiler, one step after warm-up): python scripts/diag_step_ops.py
[bench flags].  Used to find the small launches around the sweep kernels in the homography configurations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
import __graft_entry__ as ge  # noqa: E402
ge.build()
device = torch.device("cuda:0")
c = bench.make_batch(args, device, 0)
step, _ = bench.build_step(args, c, device)
for _ in range(5):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.events() if e.device_type.name == "CUDA" or str(e.device_type).endswith("CUDA")]
for e in sorted(rows, key=lambda e: e.time_range.start):
    print("%8.1f us  %s" % (e.device_time if hasattr(e, "device_time") else e.cuda_time, e.name[:110]))
print("--- CPU ops with a kernel, in order")
for e in prof.events():
    if e.device_type.name == "CPU" and (getattr(e, "device_time", 0) or getattr(e, "cuda_time", 0)) and not e.name.startswith(("hip", "cuda")):
        print("   ", e.name[:100])
