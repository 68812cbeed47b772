"""Where the host time of the small operators goes: cProfile over 300 forward+backward round trips of
smooth_loss_disp and reprojection_loss at the headline shape, device idle in between (the next_rows measurement)."""
import cProfile, pstats, sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from planedepth_amd import ops

dev = "cuda"
B, H, W = 8, 192, 640
g = torch.Generator().manual_seed(1)
img = torch.rand(B, 3, H, W, generator=g).to(dev)
disp_leaf = torch.rand(B, 1, H, W, generator=g).to(dev).requires_grad_(True)
pred_leaf = torch.rand(B, 3, H, W, generator=g).to(dev).requires_grad_(True)
g_rl = torch.randn(B, 1, H, W, generator=g).to(dev)
x0 = int(0.2 * W)

def smooth():
    ops.smooth_loss_disp(disp_leaf[..., x0:], img[..., x0:], 2.0).backward()
    disp_leaf.grad = None

def reproj():
    ops.reprojection_loss(pred_leaf, img, True).backward(g_rl)
    pred_leaf.grad = None

for name, fn in (("smooth", smooth), ("reproj", reproj)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        fn()
        torch.cuda.synchronize()
    print(name, "round trip with sync: %.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
    t0 = time.perf_counter()
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    print(name, "back to back: %.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(14)
