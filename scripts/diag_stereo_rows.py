"""Where do the per-row-shift stereo sweep and the general homography kernels differ from the fp64 oracle? (full size)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from test_gpu_parity import _mono_fullsize_case
from oracle import planedepth_oracle as orc
from planedepth_amd import ops
from planedepth_amd.synthetic import small_pose
c = _mono_fullsize_case(N_xy=49, N_xz=14, B=1, H=192, W=640, seed=500 + 640)
c["Rt"] = small_pose(None, 1, stereo=True)
cc = {k: v.double() for k, v in c.items()}
r = orc.warp_and_loss(cc["color_l"], cc["color_r"], cc["logits"], cc["sigma"], warp_type="homography_warp", distance=cc["distance"],
                      norm=cc["norm"], T=cc["Rt"], K=cc["K"], inv_K=cc["inv_K"], use_mixture_loss=True, automask=True)
exact = r["ph_map"].float()[0, 0]
grid = r["grid"]  # [BN,H,W,2] normalised
d = {k: v.cuda() for k, v in c.items()}
out = {}
for rows in (True, False):
    rgb, ph, _ = ops.plane_sweep_homography(d["color_l"], d["color_r"], d["logits"], d["sigma"], d["distance"], d["norm"], d["Rt"], d["K"],
                                            d["inv_K"], use_mixture_loss=True, automask=True, return_mean=True, stereo_rows=rows)
    out[rows] = ph.cpu()[0, 0]
for rows in (True, False):
    e = (out[rows] - exact).abs()
    m = exact.abs().max()
    print("rows" if rows else "general", "max err %.2e (norm %.2e), pixels above 1e-4*max: %d" % (e.max(), e.max() / m, int((e > 1e-4 * m).sum())))
    idx = torch.topk(e.flatten(), 8).indices
    for i in idx:
        y, x = int(i) // 640, int(i) % 640
        print("   y=%d x=%d err=%.2e exact=%.4f other=%.2e" % (y, x, e[y, x], exact[y, x], (out[not rows][y, x] - exact[y, x]).abs()))
# per-row error profile
e = (out[True] - exact).abs()
print("rows: worst rows", torch.topk(e.max(1).values, 6))
shift, mask, _ = ops.homography_matrices_fused(d["distance"], d["norm"], d["Rt"], d["K"], d["inv_K"], 2, rows=192)
ix = (grid[..., 0].reshape(1, 63, 192, 640) + 1) / 2 * 639
xs = torch.arange(640, dtype=torch.float64)
sh_exact = (ix - xs)[0, :, :, 320]
print("shift err max (px):", float((shift.cpu().double()[0] - sh_exact).abs().max()), "|shift| max", float(sh_exact.abs().max()))
