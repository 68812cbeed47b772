"""Kernel-level timing of the smaller public operators at the headline shape (run under rocprofv3 --kernel-trace --stats;
scripts/gpu_ops.sh).  Each operator runs forward + backward a few times; the profile's per-kernel averages are the result."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import planedepth_amd as pa
from planedepth_amd import ops, layers
from planedepth_amd.synthetic import dataset_intrinsics, small_pose

dev = torch.device("cuda")
B, N, H, W = 8, 49, 192, 640
g = torch.Generator().manual_seed(5)
rnd = lambda *s: torch.rand(*s, generator=g).to(dev)  # noqa: E731
K, inv_K = dataset_intrinsics(B, H, W)
K, inv_K = K.to(dev), inv_K.to(dev)
T = small_pose(g, B).to(dev)
img = rnd(B, 3, H, W).requires_grad_(True)
depth = (rnd(B, 1, H, W) * 20 + 1).requires_grad_(True)
for it in range(6):
    # depth-based warp (pred_self_images): backproject -> project -> grid_sample (border)
    cam = layers.BackprojectDepth(H, W)(depth, inv_K)
    pix = layers.Project3D(H, W)(cam, K, T)
    warped = ops.grid_sample(img, pix, padding_mode="border")
    warped.sum().backward()
    img.grad = depth.grad = None
    # standalone zero-padded sampling
    grid = (rnd(B, H, W, 2) * 2.2 - 1.1).requires_grad_(True)
    ops.grid_sample(img, grid, padding_mode="zeros").sum().backward()
    img.grad = None
    # HomographyWarp module (63 planes)
    d = (rnd(B, N) * 5 + 1).requires_grad_(True)
    n = torch.tensor([0.0, 0.0, 1.0], device=dev)[None, None].expand(B, N, 3)
    Tn = T[:, None].expand(B, N, 4, 4).reshape(B * N, 4, 4)
    Kn, iKn = K[:, None].expand(B, N, 4, 4).reshape(B * N, 4, 4), inv_K[:, None].expand(B, N, 4, 4).reshape(B * N, 4, 4)
    pc, pm = layers.HomographyWarp(H, W)(d, n, Tn, Kn, iKn)
    pc.sum().backward()
    # standalone mixture NLL
    err = rnd(B, N, H, W).requires_grad_(True)
    sg = (rnd(B, N, H, W) * 0.9 + 0.05).requires_grad_(True)
    pi = torch.softmax(rnd(B, N, H, W), 1).requires_grad_(True)
    layers.multimodal_loss(err, sg, pi, dist="laplacian").sum().backward()
    # SSIM module
    x = rnd(B, 3, H, W).requires_grad_(True)
    layers.SSIM()(x, img.detach()).sum().backward()
    # batch doubling
    ops.cat_flip(img.detach(), img.detach())
torch.cuda.synchronize()
print("done")
