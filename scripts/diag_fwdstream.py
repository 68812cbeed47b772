"""Where does the segment-stream forward differ from the plane-group forward?  python scripts/diag_fwdstream.py B N H W sign"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from planedepth_amd import _capi as C
B, N, H, W, sign = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
lib = C.load(); dev = torch.device("cuda")
g = torch.Generator().manual_seed(B * 1000 + W + N)
mk = lambda *sh: torch.rand(*sh, generator=g).to(dev)
src, tgt = mk(B, 3, H, W), mk(B, 3, H, W)
logits = (torch.randn(B, N, H, W, generator=g) * 2.0).to(dev)
sigma = (torch.rand(B, N, H, W, generator=g) * 1.2).to(dev)
disp = (300.0 * (2.0 / 300.0) ** ((torch.arange(N, dtype=torch.float32)[None] + torch.rand(B, N, generator=g) - 0.5) / max(N - 1, 1))) * (W / 640.0)
plane = disp.contiguous().to(dev)
print("disp[0]:", disp[0].tolist())
out = {}
for impl in (C.PD_IMPL_AUTO, C.PD_IMPL_ROWS1):
    d = C.SweepDesc(B, N, H, W, C.PD_WARP_DISP, C.PD_MIXTURE, sign, impl)
    k = lib.pd_sweep_stash_floats(ctypes.byref(d)) // (H * W)
    rgb = torch.full((B, 3, H, W), float("nan"), device=dev); ph = torch.full((B, 1, H, W), float("nan"), device=dev)
    stash = torch.full((B, k, H, W), float("nan"), device=dev); phm = torch.zeros(1, device=dev)
    C.check(lib.pd_plane_sweep_fwd(ctypes.byref(d), C.ptr(src), C.ptr(tgt), C.ptr(logits), C.ptr(sigma), C.ptr(plane), None, None, None, None,
                                   C.ptr(rgb), C.ptr(ph), C.ptr(phm), C.ptr(stash), C.stream_handle(dev)), "fwd")
    torch.cuda.synchronize(); out[impl] = (rgb.cpu(), stash.cpu())
d_ = (out[0][0] - out[4][0]).abs().amax(1)   # [B,H,W]
bad = (d_ > 1e-5).nonzero()
print("mismatching pixels:", bad.shape[0], "of", d_.numel())
if bad.shape[0]:
    xs = bad[:, 2]; print("x range", int(xs.min()), int(xs.max()), "rows", sorted(set(bad[:, 1].tolist())), "first", bad[:10].tolist())
    import collections; print("x histogram (by 16):", sorted(collections.Counter((xs // 16 * 16).tolist()).items()))
    d2 = (out[0][1][:, 0] - out[4][1][:, 0]).abs(); print("lse2 max diff", float(d2.max()))
