"""Every bench configuration at full size with LDS and uninitialised buffers poisoned (PD_DEBUG_POISON_LDS / _MEM): the
loss and every gradient must stay finite.  Usage: PD_DEBUG_POISON_LDS=1 PD_DEBUG_POISON_MEM=1 python scripts/poison_check.py"""
import os, sys
os.environ.setdefault("PD_DEBUG_POISON_LDS", "1")
os.environ.setdefault("PD_DEBUG_POISON_MEM", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import __graft_entry__ as e
e.build()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
base = ["bench.py", "--no_cpu_baseline", "--no_next_rows", "--no_ddp_step"]
configs = {"headline": [], "n63_xz_automask": ["--xz_levels", "14", "--automask"], "l1": ["--no_mixture"],
           "render": ["--render_probability"], "hr": ["--batch", "2", "--height", "384", "--width", "1280"],
           "homography stereo": ["--warp_type", "homography_warp"],
           "homography pose_net": ["--warp_type", "homography_warp", "--mono_pose"],
           "homography colmap": ["--warp_type", "homography_warp", "--colmap_pose"],
           "homography three views": ["--warp_type", "homography_warp", "--mono_sides", "--automask"],
           "homography three views, render, xz": ["--warp_type", "homography_warp", "--mono_sides", "--render_probability", "--xz_levels", "14"]}
bad = 0
for name, flags in configs.items():
    sys.argv = base + flags
    args = bench.parse()
    c = bench.make_batch(args, dev, 0)
    step, leaves = bench.build_step(args, c, dev)
    for _ in range(2):
        loss = step()
    torch.cuda.synchronize()
    ok = bool(torch.isfinite(loss).all()) and all(t.grad is None or bool(torch.isfinite(t.grad).all()) for t in leaves)
    print("%-40s %s  loss %.6f" % (name, "finite" if ok else "NaN / inf !!", float(loss)))
    bad += not ok
sys.exit(1 if bad else 0)
